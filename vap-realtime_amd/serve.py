"""The path's program: what ``python rvap/vap_main/vap_main.py --vap_model ... --cpc_model ... --port_num_in 50007 --port_num_out 50008
--vap_process_rate 20 --context_len_sec 2.5 --gpu --audio_gain 1.0`` does (vap_main.py:461-530, and its bc / nod twins), for MANY dialogues and
MANY GPUs: one process per GPU, each with its own engine and native front-end (``vapx_ingest_*``).  Same argument names as the reference
plus ``--streams`` (stream slots per GPU), ``--gpus`` and ``--mode``.  GPU r listens on ``port_num_in + 2 r`` / ``port_num_out + 2 r``;
a dialogue stays on the GPU it connected to (its state lives there) — there is no collective.

    python -m vap_realtime_amd.serve --vap_model asset/vap/vap_state_dict_jp_20hz_2500msec.pt --cpc_model asset/cpc/60k_epoch4-d0f474de.pt \\
        --streams 4096 --gpus 8

``--synthetic-weights SEED`` serves seeded random weights (no checkpoint files: load tests, demos).
"""
from __future__ import annotations

import argparse
import json
import multiprocessing as mp
import signal
import time


def serve_one(rank: int, n_gpus: int, args) -> None:
    from . import checkpoints, dist_util, engine, ingest, weights as W
    dist_util.pin_rank_to_cores(rank, n_gpus)
    if args.synthetic_weights is not None:
        cpc, vap = W.synthetic_weights(args.synthetic_weights, args.vap_process_rate, args.mode or "vap")
        blob, mode = W.pack_blob(cpc, vap, args.mode or "vap"), args.mode or "vap"
    else:
        blob, hz, mode = checkpoints.import_checkpoints(args.vap_model, args.cpc_model, frame_rate=args.vap_process_rate, mode=args.mode)
    eng = engine.Engine(blob, args.vap_process_rate, args.context_len_sec, max_streams=args.streams,
                        max_batch=min(args.streams, args.max_batch), mode=mode, device_id=rank,
                        groups=2)   # two intra-tick overlap groups: ragged ticks of a few hundred streams get 20 % shorter
    srv = ingest.NativeServer(eng, port_in=args.port_num_in + 2 * rank, port_out=args.port_num_out + 2 * rank, gain=args.audio_gain,
                              max_wait_s=args.max_wait_ms * 1e-3, bind_any=args.bind_any, rx_threads=args.rx_threads, tx_threads=args.tx_threads)
    print(f"[vapx] GPU {rank}: {args.streams} stream slots, mode {mode}, {args.vap_process_rate} Hz / {args.context_len_sec} s — "
          f"input :{srv.port_in}, output :{srv.port_out}", flush=True)
    stop = {"now": False}
    signal.signal(signal.SIGTERM, lambda *_: stop.__setitem__("now", True))
    signal.signal(signal.SIGINT, lambda *_: stop.__setitem__("now", True))
    last = time.time()
    while not stop["now"]:
        time.sleep(0.2)
        if args.stats_sec > 0 and time.time() - last >= args.stats_sec:
            last = time.time()
            st = srv.stats(reset_latency_window=True)
            print(f"[vapx] GPU {rank}: " + json.dumps({k: (round(v, 3) if isinstance(v, float) else v) for k, v in st.items()}), flush=True)
    srv.close()
    eng.close()


def main(argv=None) -> None:
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--vap_model", type=str, default="../../asset/vap/vap_state_dict_jp_20hz_2500msec.pt")
    ap.add_argument("--cpc_model", type=str, default="../../asset/cpc/60k_epoch4-d0f474de.pt")
    ap.add_argument("--port_num_in", type=int, default=50007)
    ap.add_argument("--port_num_out", type=int, default=50008)
    ap.add_argument("--vap_process_rate", type=int, default=20)
    ap.add_argument("--context_len_sec", type=float, default=2.5)
    ap.add_argument("--gpu", action="store_true", help="accepted for compatibility: this engine has no CPU path")
    ap.add_argument("--audio_gain", type=float, default=1.0)
    ap.add_argument("--mode", choices=["vap", "bc", "nod"], default=None, help="head set (default: inferred from the state dict)")
    ap.add_argument("--streams", type=int, default=1, help="dialogue slots per GPU (the reference serves exactly one)")
    ap.add_argument("--max_batch", type=int, default=1024)
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--max_wait_ms", type=float, default=2.0)
    ap.add_argument("--rx_threads", type=int, default=4)
    ap.add_argument("--tx_threads", type=int, default=4)
    ap.add_argument("--bind_any", action="store_true", help="listen on 0.0.0.0 instead of 127.0.0.1")
    ap.add_argument("--stats_sec", type=float, default=10.0)
    ap.add_argument("--synthetic-weights", dest="synthetic_weights", type=int, default=None)
    args = ap.parse_args(argv)
    if args.gpus <= 1:
        serve_one(0, 1, args)
        return
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=serve_one, args=(r, args.gpus, args), daemon=False) for r in range(args.gpus)]
    for p in procs:
        p.start()
    try:
        for p in procs:
            p.join()
    except KeyboardInterrupt:
        for p in procs:
            p.terminate()


if __name__ == "__main__":
    main()
