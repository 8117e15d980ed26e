import numpy as np
M32 = np.uint64(0xFFFFFFFF)
def old(idx, key):
    x = (idx.astype(np.uint64) * np.uint64(0x9E3779B9) + np.uint64(key)) & M32
    x ^= x >> np.uint64(16); x = (x * np.uint64(0x7feb352d)) & M32; x ^= x >> np.uint64(15); x = (x * np.uint64(0x846ca68b)) & M32; x ^= x >> np.uint64(16)
    return x
def mul24(a, c):
    return ((a & np.uint64(0xFFFFFF)) * np.uint64(c)) & M32
def new(idx, key, C1=0xC2B2AF, C2=0x85EBCB, s1=16, s2=13, s3=16):
    x = (idx.astype(np.uint64) + np.uint64(key)) & M32
    x ^= x >> np.uint64(s1); x = mul24(x, C1); x ^= x >> np.uint64(s2); x = mul24(x, C2); x ^= x >> np.uint64(s3)
    return x
def new3(idx, key):
    # three rounds
    x = (idx.astype(np.uint64) + np.uint64(key)) & M32
    x ^= x >> np.uint64(16); x = mul24(x, 0xC2B2AF); x ^= x >> np.uint64(13); x = mul24(x, 0x85EBCB); x ^= x >> np.uint64(11); x = mul24(x, 0x9E3779); x ^= x >> np.uint64(16)
    return x
def report(name, f):
    rng = np.random.default_rng(1)
    worst_rate = 0; worst_corr = 0; worst_bit = 0
    for trial in range(6):
        key = int(rng.integers(0, 2**32))
        base = int(rng.integers(0, 2**24)) if trial % 2 else 0
        n = 1 << 22
        idx = np.arange(base, base + n, dtype=np.uint64)
        h = f(idx, key)
        for p in (0.1, 0.3, 0.5):
            thr = np.uint64(int(p * 2**32))
            keep = (h >= thr).astype(np.float64)
            rate = keep.mean()
            worst_rate = max(worst_rate, abs(rate - (1 - p)) / np.sqrt(p * (1 - p) / n))      # in sigmas
            k0 = keep - keep.mean()
            for lag in (1, 2, 3, 4, 8, 16, 64, 128, 197, 224, 512, 1024, 1536, 197 * 197, 12608):
                c = (k0[:-lag] * k0[lag:]).mean() / k0.var()
                worst_corr = max(worst_corr, abs(c) * np.sqrt(n))                               # in sigmas
            # 2-D: a [rows][512] site: column means and row means
            K = keep[: (n // 512) * 512].reshape(-1, 512)
            cm = K.mean(0); rm = K.mean(1)
            worst_corr = max(worst_corr, abs(cm - (1 - p)).max() / np.sqrt(p * (1 - p) / K.shape[0]) / 1.5, abs(rm - (1 - p)).max() / np.sqrt(p * (1 - p) / 512) / 1.5)
        bits = ((h[:, None] >> np.arange(32, dtype=np.uint64)[None, :]) & np.uint64(1)).astype(np.float64)
        worst_bit = max(worst_bit, np.abs(bits.mean(0)[16:] - 0.5).max() / np.sqrt(0.25 / n))
    print(f"{name:8s} keep-rate dev {worst_rate:6.2f} sigma   worst lag/col/row stat {worst_corr:6.2f} (sigma-ish)   top-16-bit bias {worst_bit:6.2f} sigma")
report("old", old)
report("new2", new)
report("new3", new3)
for C1, C2 in ((0xD3833F, 0x8DA6B3), (0xA54FF5, 0xC4CEB9)):
    report(f"n2 {C1:x}", lambda i, k: new(i, k, C1, C2))
report("n2 s15", lambda i, k: new(i, k, s1=15, s2=12, s3=15))

def score(f, trials=4):
    rng = np.random.default_rng(7)
    worst = 0
    n = 1 << 21
    for trial in range(trials):
        key = int(rng.integers(0, 2**32)); base = int(rng.integers(0, 2**24)) if trial % 2 else 0
        idx = np.arange(base, base + n, dtype=np.uint64)
        h = f(idx, key)
        for p in (0.1, 0.5):
            keep = (h >= np.uint64(int(p * 2**32))).astype(np.float64)
            worst = max(worst, abs(keep.mean() - (1 - p)) / np.sqrt(p * (1 - p) / n))
            k0 = keep - keep.mean()
            for lag in (1, 2, 3, 4, 8, 16, 64, 128, 197, 224, 512, 1024, 1536, 197 * 197, 12608):
                worst = max(worst, abs((k0[:-lag] * k0[lag:]).mean() / k0.var()) * np.sqrt(n))
            K = keep[: (n // 512) * 512].reshape(-1, 512)
            worst = max(worst, abs(K.mean(0) - (1 - p)).max() / np.sqrt(p * (1 - p) / K.shape[0]) / 1.5)
    return worst
print("old", score(old), "new3", score(new3))
rng = np.random.default_rng(3)
best = []
for it in range(40):
    C1 = int(rng.integers(1 << 22, 1 << 24)) | 1; C2 = int(rng.integers(1 << 22, 1 << 24)) | 1
    s1, s2, s3 = int(rng.integers(12, 18)), int(rng.integers(9, 16)), int(rng.integers(13, 18))
    sc = score(lambda i, k: new(i, k, C1, C2, s1, s2, s3), trials=2)
    best.append((sc, hex(C1), hex(C2), s1, s2, s3))
best.sort()
print(best[:6])
print("---- validation of the search's best on the full report (other keys, 4M samples)")
report("cand A", lambda i, k: new(i, k, 0xc318ef, 0x6b38d5, 16, 15, 15))
report("cand B", lambda i, k: new(i, k, 0x4c3ab7, 0xa46cf1, 13, 12, 16))
# attention-site pattern: idx = (ch*S + q)*S + key for S = 197: check row (fixed q) and column (fixed key) keep rates over 256 heads
def att(f, name):
    S = 197; ch = np.arange(256, dtype=np.uint64)[:, None, None]; q = np.arange(S, dtype=np.uint64)[None, :, None]; kk = np.arange(S, dtype=np.uint64)[None, None, :]
    idx = ((ch * np.uint64(S) + q) * np.uint64(S) + kk).reshape(-1)
    h = f(idx, 0x1234ABCD).reshape(256, S, S)
    keep = (h >= np.uint64(int(0.1 * 2**32))).astype(np.float64)
    n1 = 256 * S
    print(name, "attention site: overall", keep.mean(), "worst per-key column dev (sigma)", abs(keep.mean((0, 1)) - 0.9).max() / np.sqrt(0.09 / n1),
          "worst per-query row dev", abs(keep.mean((0, 2)) - 0.9).max() / np.sqrt(0.09 / n1), "worst per-head dev", abs(keep.mean((1, 2)) - 0.9).max() / np.sqrt(0.09 / (S * S)))
att(old, "old"); att(lambda i, k: new(i, k, 0xc318ef, 0x6b38d5, 16, 15, 15), "cand A")
