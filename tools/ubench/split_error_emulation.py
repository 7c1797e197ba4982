"""[r6] Error class of split formats for the acting kernel's 256 -> 512 product, emulated in numpy against fp64 (docs/LEVERS.md, Round 6): fp32 arithmetic,
the six-term exact bf16 split the kernels run, and a two-way fp16 split (three / four partial products) that was evaluated and NOT built.  CPU only."""
import numpy as np
rng = np.random.default_rng(0)
R, K, N = 4096, 256, 512
# h1 = relu(layernorm(z)) style activations, W2 uniform(-sqrt(6/256), +)
z = rng.normal(0, 1, (R, K)).astype(np.float32)
h1 = np.maximum(z, 0).astype(np.float32)
W = rng.uniform(-np.sqrt(6/256), np.sqrt(6/256), (N, K)).astype(np.float32)
ref = h1.astype(np.float64) @ W.astype(np.float64).T

def bf16(x):
    u = x.astype(np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32)
def split_bf16(x):
    hi = bf16(x); r = (x - hi).astype(np.float32); mid = bf16(r); lo = bf16((r - mid).astype(np.float32)); return hi, mid, lo
def split_f16(x):
    hi = x.astype(np.float16).astype(np.float32); lo = (x - hi).astype(np.float16).astype(np.float32); return hi, lo
def dot32(a, b):
    # fp32 accumulation in k-chunks of 32 like an MFMA chain: emulate as float32 matmul per 32-chunk, summed sequentially in fp32
    out = np.zeros((a.shape[0], b.shape[0]), np.float32)
    for k in range(0, K, 32):
        out = (out + (a[:, k:k+32].astype(np.float64) @ b[:, k:k+32].astype(np.float64).T).astype(np.float32)).astype(np.float32)
    return out
def seq32(a, b):
    out = np.zeros((a.shape[0], b.shape[0]), np.float32)
    for k in range(0, K, 4):
        out = (out + (a[:, k:k+4].astype(np.float64) @ b[:, k:k+4].astype(np.float64).T).astype(np.float32)).astype(np.float32)
    return out
def report(name, got):
    e = np.abs(got.astype(np.float64) - ref)
    print(f"{name:34s} max {e.max():.3e} mean {e.mean():.3e}")
report("fp32 mfma-like (k=4 chunks)", seq32(h1, W))
ah, am, al = split_bf16(h1); bh, bm, bl = split_bf16(W)
rest = np.zeros((R, N), np.float32)
for (a, b) in ((al, bh), (ah, bl), (am, bm), (am, bh), (ah, bm)):
    rest = (rest + dot32(a, b)).astype(np.float32)
report("bf16 x6 split", (dot32(ah, bh) + rest).astype(np.float32))
fh, fl = split_f16(h1); gh, gl = split_f16(W)
rest = (dot32(fl, gh) + dot32(fh, gl)).astype(np.float32)
report("fp16 x3 split (hh, hl, lh)", (dot32(fh, gh) + rest).astype(np.float32))
rest4 = (rest + dot32(fl, gl)).astype(np.float32)
report("fp16 x4 split (+ll)", (dot32(fh, gh) + rest4).astype(np.float32))
# representation error alone
print("h1 repr err fp16x2 max rel", np.max(np.abs((fh.astype(np.float64)+fl) - h1) / np.maximum(np.abs(h1), 1e-30)[...]))
print("W repr err fp16x2 max abs", np.max(np.abs((gh.astype(np.float64)+gl) - W)))
