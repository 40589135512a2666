"""Model-based random test of the C ABI: a seeded random program of the operations a many-stream server performs — steps over
random subsets of the streams in random batch order (what ragged ticks of the TCP front-end look like), stream resets, carry-only
resets (reconnects), state export / import into another slot, host and device paths, overlap groups — runs against the engine while
one oracle instance per dialogue follows the same program on the CPU.  Every stepped stream must agree with its oracle (<= 1e-4)
at every tick, through window fill and slide — on the default fp32 path and on the split-precision path (VAPX_FLAG_SPLIT_F16).  Complements
the fixed scenarios of test_engine_gpu.py."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL = 1e-4


class Dialogue:
    """One stream's oracle state + its server-side carry."""

    def __init__(self, oracle, hop):
        from oracle.vap_oracle import ServerFramer
        self.o, self.hop = oracle, hop
        self.st, self.fr = oracle.new_state(1), ServerFramer(1, hop)

    def step(self, new):                      # new: [2, hop]
        return self.o.step(self.fr.frame(new[None]), self.st)

    def reset(self, carry_only=False):
        from oracle.vap_oracle import ServerFramer
        self.fr = ServerFramer(1, self.hop)
        if not carry_only:
            self.st = self.o.new_state(1)


def _compare(mode, got, i, want):
    """max |hip - oracle| over the outputs the `mode` program publishes, for batch slot i of `got` and the 1-stream oracle result."""
    pairs = [(got["vad"][i], want["vad"][0])]
    if mode == "vap":
        pairs += [(got[k][i], want[k][0]) for k in ("p_now", "p_future", "logits")]
    elif mode == "bc":
        pairs += [(got["aux"][i, 1], want["p_bc_react"][0]), (got["aux"][i, 2], want["p_bc_emo"][0])]
    else:
        pairs += [(got["aux"][i, 1], want["p_nod_short"][0]), (got["aux"][i, 2], want["p_nod_long"][0]), (got["aux"][i, 3], want["p_nod_long_p"][0])]
        n = want["p_bc"].shape[1]                       # p_bc of every window row (the reference's batch-index quirk)
        pairs.append((got["logits"][i, :n], want["p_bc"][0]))
    return max(float(np.abs(np.asarray(a, np.float64).reshape(-1) - np.asarray(b, np.float64).reshape(-1)).max()) for a, b in pairs)


@pytest.mark.parametrize("split", [False, True], ids=["fp32", "split_f16"])
@pytest.mark.parametrize("seed,hz,ctx,groups,mode", [(1, 20, 2.5, 0, "vap"), (2, 20, 1.0, 2, "vap"), (3, 50, 1.3, 2, "vap"), (4, 10, 2.5, 2, "vap"),
                                                     (5, 20, 2.5, 0, "nod"), (6, 10, 5.0, 2, "bc"), (7, 5, 10.0, 0, "vap"), (8, 50, 5.0, 0, "vap")])
def test_random_program_against_per_stream_oracles(seed, hz, ctx, groups, mode, split):
    import torch
    from oracle.vap_oracle import VapOracle
    from vap_realtime_amd import engine, synth, weights as W
    rng = np.random.default_rng(seed)
    cpc, vap = W.synthetic_weights(30 + seed, hz, mode)
    oracle = VapOracle(cpc, vap, hz, ctx, mode=mode)
    hop = 16000 // hz
    S, slots, ticks = 5, 9, int(ctx * hz) + 14          # 5 dialogues living in 9 engine slots
    audio = synth.dialogue_batch([70 + i for i in range(S)], hop * ticks)
    eng = engine.Engine(W.pack_blob(cpc, vap, mode), hz, ctx, max_streams=slots, max_batch=slots, groups=groups, mode=mode, split_f16=split)
    dia = [Dialogue(oracle, hop) for _ in range(S)]
    slot_of = list(rng.permutation(slots)[:S])            # dialogue k lives in engine slot slot_of[k]
    pos = [0] * S                                         # next audio frame of each dialogue
    d_out = torch.zeros(slots, engine.OUT_STRIDE, device="cuda")
    worst, steps, n_ops = 0.0, 0, {"reset": 0, "carry": 0, "migrate": 0, "device": 0}
    for t in range(ticks):
        op = rng.random()
        if op < 0.06:                                     # full reset of one dialogue
            k = int(rng.integers(S)); dia[k].reset(); eng.reset_stream(int(slot_of[k])); n_ops["reset"] += 1
        elif op < 0.12:                                   # reconnect: only the carry restarts (vap_main.py:368-369)
            k = int(rng.integers(S)); dia[k].reset(carry_only=True); eng.reset_carry(int(slot_of[k])); n_ops["carry"] += 1
        elif op < 0.20:                                   # migrate a dialogue to a free slot through get_state / set_state
            k = int(rng.integers(S))
            free = [s for s in range(slots) if s not in slot_of]
            dst = int(rng.choice(free))
            state = eng.get_state(int(slot_of[k]))
            eng.reset_stream(int(slot_of[k]))
            eng.set_state(dst, state)
            slot_of[k] = dst; n_ops["migrate"] += 1
        members = [k for k in range(S) if rng.random() < 0.75 and pos[k] < ticks]
        if not members:
            continue
        order = list(rng.permutation(members))
        new = np.stack([audio[k, :, pos[k] * hop:(pos[k] + 1) * hop] for k in order]).astype(np.float32)
        ids = np.array([slot_of[k] for k in order], dtype=np.int32)
        if rng.random() < 0.3:                            # device path (what the bench times)
            d_audio = torch.from_numpy(np.ascontiguousarray(new)).cuda()
            d_ids = torch.from_numpy(ids).cuda()
            eng.step_device(len(order), d_audio.data_ptr(), hop, d_out.data_ptr(), ids_ptr=d_ids.data_ptr(), stream=0,
                            defer_join=bool(groups) and rng.random() < 0.5)
            eng.join(0)
            torch.cuda.synchronize()
            got = engine.split_outputs(d_out[:len(order)].cpu().numpy())
            n_ops["device"] += 1
        else:
            got = engine.split_outputs(eng.step(new, ids))
        for i, k in enumerate(order):
            want = dia[k].step(audio[k, :, pos[k] * hop:(pos[k] + 1) * hop])
            pos[k] += 1
            d = _compare(mode, got, i, want)
            worst = max(worst, d)
            assert d <= TOL, f"tick {t} dialogue {k} (slot {slot_of[k]}): |hip - oracle| = {d:.3e}; ops so far {n_ops}"
            steps += 1
    eng.close()
    print(f"seed {seed}: {steps} stream-steps, ops {n_ops}, worst |hip - oracle| = {worst:.2e}")
    assert steps > 2 * ticks
