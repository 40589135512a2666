import sys, ctypes as C, numpy as np
sys.path.insert(0, '/root/repo')
import torch
from vap_realtime_amd import engine, synth, weights as W
S = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
cpc, vap = W.synthetic_weights(0, 20)
eng = engine.Engine(W.pack_blob(cpc, vap), 20, 2.5, max_streams=S)
a = synth.noise_batch(S, 800 * 4)
d_a = torch.from_numpy(a).cuda(); d_out = torch.zeros(S, engine.OUT_STRIDE, device='cuda')
for i in range(54):
    x = d_a[:, :, (i % 4) * 800:(i % 4 + 1) * 800].contiguous()
    eng.step_device(S, x.data_ptr(), 800, d_out.data_ptr())
torch.cuda.synchronize()
nb = min(S * 2, 8192)
buf = np.zeros(nb * 4 * 12, np.uint64)
lib = engine.load_library()
lib.vapx_debug_stamps.argtypes = [C.c_void_p, C.c_size_t]
rc = lib.vapx_debug_stamps(buf.ctypes.data_as(C.c_void_p), buf.size); assert rc == 0
st = buf.reshape(nb, 4, 12).astype(np.int64)
# last launch = cross block of layer 2 (no qx): stamps 0..7 + 9(end)
d = np.diff(st[:, :, :9], axis=2)
valid = st[:, :, 8] != 0
print("last attn launch (cross block), mean cycles per phase [load+Vstage, tile0, tile1, barrier, sAtt, proj mm(+resid), LN stats, normalise+stores]:")
print(np.round(d.reshape(-1, 8).mean(0)).astype(int), "total", int((st[:, :, 8 if not valid.any() else 9] - st[:, :, 0]).mean()))
print("per-wave (head) totals:", [(int((st[:, h, 7] - st[:, h, 0]).mean())) for h in range(4)])
print("clock: readcyclecounter units")
