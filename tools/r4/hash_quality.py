"""Statistical quality of csrc/attn2.hip's pair_hash (two 24-bit multiply rounds; one 32-bit word decides the dropout of two adjacent
keys).  CPU only (numpy).  Prints, per seed: the drop rate of both 16-bit fields at p = 0.1, the correlation between the two fields of
a word, the autocorrelation of the keep mask at the strides an attention tile walks (neighbouring pairs, a 416-key row = 208 pairs,
powers of two), and the spread of column / row means of a [rows, 208] mask against the binomial expectation.
Run: python tools/r4/hash_quality.py  (output kept in profiles/r04_hash_quality.txt)"""
import numpy as np

M1, M2 = 0x9E3779, 0x85EBCB
FOLD_HIGH_BITS = True      # round 5: a ^= a >> 12 before the first 24-bit multiply


def pair_hash(idx, s0, s1):
    idx = idx.astype(np.uint64)
    a = (idx ^ s0) & 0xFFFFFFFF
    if FOLD_HIGH_BITS:
        a ^= a >> 12
    h = ((a & 0xFFFFFF) * M1 + s1) & 0xFFFFFFFF
    h ^= h >> 15
    h = ((h & 0xFFFFFF) * M2 + (a >> 8)) & 0xFFFFFFFF
    h ^= h >> 13
    return h


def main():
    p = 0.1
    t = int(p * 65536.0 + 0.5)
    for seed in (0, 1, 12345, 0xDEADBEEFCAFE, 0x1234567 + 416):
        s0 = seed & 0xFFFFFFFF
        s1 = ((seed >> 32) ^ ((seed & 0xFFFFFFFF) * 0x9E3779B9)) & 0xFFFFFFFF
        n = 1 << 22
        h = pair_hash(np.arange(n, dtype=np.uint64), s0, s1)
        lo, hi = h & 0xFFFF, h >> 16
        klo, khi = lo >= t, hi >= t
        strides = (1, 2, 104, 208, 256, 4096, 65536)
        cs = [np.corrcoef(klo[:-s], klo[s:])[0, 1] for s in strides]
        print(f"seed {seed:#x}: drop rate field0 {1 - klo.mean():.5f} field1 {1 - khi.mean():.5f} (p = {p}); corr(field0, field1) {np.corrcoef(klo, khi)[0, 1]:+.4f}")
        print("    autocorrelation of the keep mask at pair strides " + ", ".join(f"{s}: {c:+.4f}" for s, c in zip(strides, cs)))
        # pairs 2^24 apart (B * H * Sq * round8(Sk) / 2 reaches 2^24 at B = 8, H = 8, S ~ 1066): the first multiply only sees 24 bits of the index
        base = np.arange(1 << 20, dtype=np.uint64)
        far = pair_hash(base + (1 << 24), s0, s1)
        near = pair_hash(base, s0, s1)
        kf, kn = (far & 0xFFFF) >= t, (near & 0xFFFF) >= t
        print(f"    stride 2^24: mask agreement {np.mean(kf == kn):.4f} (independent masks: {1 - 2 * p * (1 - p):.4f}), correlation {np.corrcoef(kf, kn)[0, 1]:+.4f}")
        m = klo[:208 * 20000].reshape(20000, 208)
        print(f"    [20000 x 208] mask: column-mean std {m.mean(0).std():.5f} (binomial {np.sqrt(p * (1 - p) / 20000):.5f}), "
              f"row-mean std {m.mean(1).std():.5f} (binomial {np.sqrt(p * (1 - p) / 208):.5f})")


if __name__ == "__main__":
    main()
