#!/bin/bash
# One command for the day a Swift 6.2 toolchain and an MI355X meet: builds libhe_amd.so, builds this package against an
# UNMODIFIED checkout of apple/swift-homomorphic-encryption (no clone: the checkout is given), and runs the package's tests
# -- the parity tests of Tests/HeAmdTests/HeAmdTests.swift and the reference's own generic suites instantiated over the
# drop-ins (Tests/HeAmdTests/ReferenceSuites.swift: HeAPITests.swift:92-113 over GpuBfv, IndexPirTests.swift:23-137 over
# MulPirServer<GpuPirUtil<...>>).
#
#   swift/ci.sh [/path/to/swift-homomorphic-encryption]        (default: $SWIFT_HE_PATH, then /root/reference)
#
# Exit status: 0 = everything built and every test passed; 2 = a prerequisite is missing (named on stderr, nothing built).
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
ROOT="$(dirname "$HERE")"
REFERENCE="${1:-${SWIFT_HE_PATH:-/root/reference}}"

missing=0
need() { echo "swift/ci.sh: $1" >&2; missing=1; }
command -v swift >/dev/null 2>&1 || need "no swift on PATH (the reference needs swift-tools-version 6.2: Package.swift:1)"
if command -v swift >/dev/null 2>&1; then
  version="$(swift --version 2>/dev/null | sed -n 's/.*Swift version \([0-9]*\.[0-9]*\).*/\1/p' | head -1)"
  case "$version" in
    6.[2-9]*|[7-9].*) ;;
    *) need "swift $version found, 6.2 or later needed" ;;
  esac
fi
command -v hipcc >/dev/null 2>&1 || [ -x /opt/rocm/bin/hipcc ] || need "no hipcc (ROCm) to build libhe_amd.so"
[ -f "$REFERENCE/Package.swift" ] || need "no reference checkout at $REFERENCE (pass its path, or set SWIFT_HE_PATH)"
if command -v rocminfo >/dev/null 2>&1; then
  rocminfo 2>/dev/null | grep -q gfx950 || need "no gfx950 (MI355X) device visible to rocminfo: the tests run kernels"
else
  need "no rocminfo: cannot tell whether an MI355X is present"
fi
[ "$missing" -eq 0 ] || exit 2

# 1. the library (hipcc --offload-arch=gfx950; in-tree, nothing installed)
python3 "$ROOT/swift-homomorphic-encryption_amd/build.py"
export HE_AMD_LIB_DIR="$ROOT/swift-homomorphic-encryption_amd/lib"
export SWIFT_HE_PATH="$REFERENCE"
export LD_LIBRARY_PATH="$HE_AMD_LIB_DIR:${LD_LIBRARY_PATH:-}"
# the header the C target ships must be the product header (tests/test_swift_package.py holds them equal)
cmp "$ROOT/include/he_amd.h" "$HERE/Sources/CHeAmd/include/he_amd.h"

# 2. the package, with the reference's own settings (README.md:99-112 of the reference: cross-module optimisation is in
#    Package.swift's swiftSettings already)
cd "$HERE"
swift build -c release
# 3. the tests: parity through the Swift API, then the reference's suites over the drop-ins
swift test -c release --filter HeAmdTests
echo "swift/ci.sh: built against $REFERENCE and every test passed"
