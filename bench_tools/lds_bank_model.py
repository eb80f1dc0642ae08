"""LDS bank-conflict model of the NTT transposes (no GPU needed).

Applies the gfx950 banking rules of MI355X_MICROARCH.md ("LDS [CDNA4]") to the padded tile addressing of
csrc/ntt_common.hpp and prints, per pass of a (LOGN, LOGT) kernel, the LDS-array cycles of one wave's store and
load instruction (ideal: ds_read_b64 2, ds_write_b64 4).

  ds_read_b64 : lane groups {0-31}, {32-63}; bank = (byte/4) mod 64  -> 8-byte word slot mod 32 within a group
  ds_write_b64: 4 groups of 16 contiguous lanes; bank = (byte/4) mod 32 -> word slot mod 16 within a group
Each extra distinct address on a busy bank adds one cycle for that group.

Usage: python bench_tools/lds_bank_model.py [LOGN LOGT] [pad-expression]
"""
import sys


def default_slot(idx):
    return idx + (idx >> 3) + ((idx >> 8) << 3)


def element_index(logn, loge, lo, w, r, tid):
    if w == loge:
        return ((tid >> lo) << (lo + loge)) | (r << lo) | (tid & ((1 << lo) - 1))
    x = loge - w
    wb = max(logn - loge - 6, 0)
    wave, lane = tid >> 6, tid & 63
    return (wave << (logn - wb)) | ((r >> w) << (logn - wb - x)) | (lane << w) | (r & ((1 << w) - 1))


def group_cycles(slots, banks):
    per_bank = {}
    for s in slots:
        per_bank.setdefault(s % banks, set()).add(s)
    return max(len(v) for v in per_bank.values())


def wave_cycles(slots64, kind):
    if kind == "read":
        return sum(group_cycles(slots64[g * 32:(g + 1) * 32], 32) for g in range(2))
    return sum(group_cycles(slots64[g * 16:(g + 1) * 16], 16) for g in range(4))


def passes(logn, loge):
    p = (logn + loge - 1) // loge
    r = logn - (p - 1) * loge
    fwd = [(logn - (k + 1) * loge, loge) for k in range(p - 1)] + [(0, r)]
    return fwd


def report(logn, logt, slot):
    loge = logn - logt
    waves = (1 << logt) // 64
    print(f"N=2^{logn}, {1 << logt} lanes, {1 << loge} words/lane; passes (LO, W): {passes(logn, loge)}")
    for lo, w in passes(logn, loge):
        for kind in ("write", "read"):
            worst, total = 0, 0
            for wave in range(waves):
                for r in range(1 << loge):
                    slots = [slot(element_index(logn, loge, lo, w, r, wave * 64 + lane)) for lane in range(64)]
                    c = wave_cycles(slots, kind)
                    worst = max(worst, c)
                    total += c
            ideal = 2 if kind == "read" else 4
            n = waves * (1 << loge)
            print(f"  pass LO={lo:2d} W={w}: {kind:5s} avg {total / n:5.2f} worst {worst:2d} cycles/instr (ideal {ideal})")


if __name__ == "__main__":
    args = sys.argv[1:]
    slot = default_slot
    if len(args) >= 3:
        slot = eval("lambda idx: " + args[2])
    if len(args) >= 2:
        report(int(args[0]), int(args[1]), slot)
    else:
        for logn, logt in ((12, 9), (13, 9), (13, 10), (14, 10)):
            report(logn, logt, slot)
