import sys, ctypes as C, numpy as np
sys.path.insert(0, '/root/repo')
import torch
from vap_realtime_amd import engine, synth, weights as W
S = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
cpc, vap = W.synthetic_weights(0, 20)
eng = engine.Engine(W.pack_blob(cpc, vap), 20, 2.5, max_streams=S, full_last_layer=True)
a = synth.noise_batch(S, 800 * 4)
d_a = torch.from_numpy(a).cuda(); d_out = torch.zeros(S, engine.OUT_STRIDE, device='cuda')
for i in range(54):
    x = d_a[:, :, (i % 4) * 800:(i % 4 + 1) * 800].contiguous()
    eng.step_device(S, x.data_ptr(), 800, d_out.data_ptr())
torch.cuda.synchronize()
nb = min(S * 2 * 50 // 32, 16384)
buf = np.zeros(16384 * 4 * 24, np.uint64)
lib = engine.load_library()
lib.vapx_debug_stamps.argtypes = [C.c_void_p, C.c_size_t]
assert lib.vapx_debug_stamps(buf.ctypes.data_as(C.c_void_p), buf.size) == 0
st = buf.reshape(16384, 4, 24).astype(np.int64)[:nb]
cnt = (st != 0).sum(axis=2)
print("stamps per wave:", np.unique(cnt))
d = np.diff(st, axis=2)
names = ["tile+LN", "ffn1.0", "gelu0", "sync0", "ffn2.0", "ffn1.1", "gelu1", "sync1", "ffn2.1", "ffn1.2", "gelu2", "sync2", "ffn2.2", "resid+st", "kvx", "LN", "end"]
m = d.reshape(-1, 23).mean(0)
k = int(cnt.max()) - 1
for n_, v in zip(names, m[:k]): print(f"{n_:10s} {v:9.0f}")
print("total", (st[:, :, k] - st[:, :, 0]).mean())
