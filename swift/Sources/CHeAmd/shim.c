// CHeAmd has no code of its own: it exposes include/he_amd.h of libhe_amd.so to Swift, the way CUtil exposes zeroize.h
// (reference Package.swift:100-105).  SwiftPM wants at least one translation unit per C target.
#include "he_amd.h"

const char* che_amd_header_version(void) { return "he_amd.h (C ABI of libhe_amd.so)"; }
