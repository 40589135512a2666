"""Weight packing of the row-stationary split-precision block experiment (tools/microbench/ffn_rows_f16x3.hip; round 5, not shipped).
Kept for the record: at commit 78eb083 these functions lived in vap-realtime_amd/weights.py, the kernel was wired into the engine
(blob entries L<l>.wrs) and passed tests/test_split_precision_gpu.py.  `layer_program` is what pack_blob appended per layer."""
import numpy as np

F16X3_WEIGHT_SCALE = 256.0
F16X3_SAT = 255.0 * 256.0


def frag_pack_f16x3_rs(W: np.ndarray, n0: int, k0: int, halves: bool = False) -> np.ndarray:
    """256x256 sub-matrix -> the weight stream of the ROW-STATIONARY split-precision block (csrc/ffn_rows_f16x3.hip): a wave keeps its 16
    rows in registers through the whole chain and reads every weight fragment (the A operand of v_mfma_f32_16x16x32_f16: 16 output
    columns x 32 k) from an LDS ring that LDS-DMA fills linearly, so the stream is laid out in exactly the order it is consumed.
    w' = 2^8 w = hi + lo (f16).
    Full order  [8 kc][16 t][2 hi/lo][64 lane][8 s]  (one k-step of 32 = all sixteen 16-column tiles, 32 KB);
    halves      [2 g][8 kc][8 t'][2 hi/lo][64 lane][8 s], t = 8 g + t'  (the 128-column halves one after the other, 16 KB per k-step).
    value = w'[n0 + 16 t + (lane & 15)][k0 + 32 kc + 16 (s >> 2) + 4 (lane >> 4) + (s & 3)]: the k order inside a 32-chunk is the one in
    which a lane's accumulator registers of the PREVIOUS contraction hold its row (register r of tile t <-> column 16 t + 4 (lane >> 4) + r;
    a k-chunk = two tiles), so accumulators become the next B operand without moving."""
    sub = np.ascontiguousarray(W[n0:n0 + 256, k0:k0 + 256], dtype=np.float32) * np.float32(F16X3_WEIGHT_SCALE)
    sub = np.clip(sub, -F16X3_SAT, F16X3_SAT)
    hi = sub.astype(np.float16)
    lo = (sub - hi.astype(np.float32)).astype(np.float16)
    kc, g, s_ = np.meshgrid(np.arange(8), np.arange(4), np.arange(8), indexing="ij")
    kidx = 32 * kc + 16 * (s_ >> 2) + 4 * g + (s_ & 3)                                   # [kc][g][s]

    def lay(a):   # [n = (t, i)][k] -> [kc][t][g][i][s]
        x = a.reshape(16, 16, 256)[:, :, kidx]                                           # [t][i][kc][g][s]
        return x.transpose(2, 0, 3, 1, 4)
    both = np.stack([lay(hi), lay(lo)], axis=2)                                          # [kc][t][hl][g][i][s]
    if halves:
        both = both.reshape(8, 2, 8, 2, 4, 16, 8).transpose(1, 0, 2, 3, 4, 5, 6)         # [g2][kc][t'][hl][g][i][s]
    return np.ascontiguousarray(both).reshape(-1).view(np.float32)



def layer_program(vap_sd, l, A):
    """The layer's weight PROGRAM in consumption order: output projection (self for layer 0, cross otherwise), per 128-wide hidden
    sub-chunk [W0 rows | W3 columns], then the NEXT layer's cross K, cross V, Q, K, V as 128-column halves (layers 0-2)."""
    def lsrc(l):
        return "ar_channel.layers.0" if l == 0 else f"ar.layers.{l - 1}"
    src = lsrc(l)
    w0, w3 = A(vap_sd[f"{src}.ffnetwork.0.weight"]), A(vap_sd[f"{src}.ffnetwork.3.weight"])
    prog = [frag_pack_f16x3_rs(A(vap_sd[f"{src}.mha.proj.weight" if l == 0 else f"{src}.mha_cross.proj.weight"]), 0, 0)]
    for c in range(3):
        a0 = frag_pack_f16x3_rs(w0, c * 256, 0, halves=True)
        a3 = frag_pack_f16x3_rs(w3, 0, c * 256)
        prog += [a0[:32768], a3[:32768], a0[32768:], a3[32768:]]
    if l < 3:
        nx = lsrc(l + 1)
        for nm in ("mha_cross.key", "mha_cross.value", "mha.query", "mha.key", "mha.value"):
            prog.append(frag_pack_f16x3_rs(A(vap_sd[f"{nx}.{nm}.weight"]), 0, 0, halves=True))
    return np.concatenate(prog)
