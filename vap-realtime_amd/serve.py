"""The path's program: what ``python rvap/vap_main/vap_main.py --vap_model ... --cpc_model ... --port_num_in 50007 --port_num_out 50008
--vap_process_rate 20 --context_len_sec 2.5 --gpu --audio_gain 1.0`` does (vap_main.py:461-530, and its bc / nod twins), for MANY dialogues and
MANY GPUs behind the reference's ONE port pair.  One process: one engine per GPU (``vapx_create`` with ``device_id = r``), one passive native
front-end per engine (its own receive / tick / send threads, ``vapx_ingest_*``), and one front door (``vapx_frontdoor_*``) that owns
``port_num_in`` / ``port_num_out`` and hands every accepted connection to a GPU.  Dialogue k lands on GPU ``k mod N`` (lowest free global slot) and
stays there — its state lives there; there is no collective.  Same argument names as the reference plus ``--streams`` (slots per GPU), ``--gpus``,
``--mode`` and ``--precision {auto,fp32,split}``.  ``auto`` (default) serves the arithmetic the load needs: fp32 — the reference's — while
``streams x frame rate`` keeps one GPU's fp32 path <= 85 % busy, the fp32-accurate split-precision engine beyond that, and it says so; when
NEITHER path holds the 10 ms bound at the requested ``--streams`` it refuses to start (``--allow-overload`` downgrades that to a warning).  The
rule and its measured rates: ``capacity.plan`` / DESIGN.md §5.

    python -m vap_realtime_amd.serve --vap_model asset/vap/vap_state_dict_jp_20hz_2500msec.pt --cpc_model asset/cpc/60k_epoch4-d0f474de.pt \\
        --streams 4096 --gpus 8

``--worker-procs`` (``auto`` turns it on when one process could not hold the descriptors: two sockets per dialogue against RLIMIT_NOFILE): the
front door becomes a process of its own that only accepts and routes, and every GPU gets a WORKER process — engine, front-end threads, sockets —
to which accepted connections are passed (``vapx_frontdoor_open_links`` / ``vapx_ingest_attach_link``, include/vapx.h).  Same ports, same
placement, same packets; 8 x 4096 dialogues need it in a container whose descriptor limit is 20 000.

``--synthetic-weights SEED`` serves seeded random weights (no checkpoint files: load tests, demos).  SIGTERM / SIGINT stop every GPU's
front-end and engine in order; a failure while one GPU comes up tears the others down and exits non-zero.
"""
from __future__ import annotations

import argparse
import json
import signal
import sys
import time


def load_blob(args):
    from . import checkpoints, weights as W
    if args.synthetic_weights is not None:
        cpc, vap = W.synthetic_weights(args.synthetic_weights, args.vap_process_rate, args.mode or "vap")
        return W.pack_blob(cpc, vap, args.mode or "vap"), args.mode or "vap"
    blob, hz, mode = checkpoints.import_checkpoints(args.vap_model, args.cpc_model, frame_rate=args.vap_process_rate, mode=args.mode)
    return blob, mode


def choose_precision(args, mode):
    if args.precision == "auto":                 # serve the arithmetic the load needs (capacity.plan: measured sustained rate per path)
        from . import capacity
        pl = capacity.plan(args.streams, args.vap_process_rate, args.context_len_sec, mode)
        args.precision_plan = pl
        print(f"[vapx] --precision auto -> {pl['precision']}: {pl['reason']}", file=sys.stderr, flush=True)
        if not pl["ok"]:
            if not args.allow_overload:
                raise RuntimeError(pl["reason"] + " (start anyway with --allow-overload, or name a --precision)")
            print("[vapx] WARNING: starting overloaded (--allow-overload): frames WILL be answered later than 10 ms at full occupancy", file=sys.stderr, flush=True)
        args.precision = pl["precision"]


def wants_worker_procs(args) -> bool:
    """``--worker-procs auto``: a process per GPU when ONE process could not hold the sockets (2 per dialogue + slack) under RLIMIT_NOFILE."""
    if args.worker_procs != "auto":
        return args.worker_procs == "on"
    import resource
    hard = resource.getrlimit(resource.RLIMIT_NOFILE)[1]
    need = 2 * max(1, args.gpus) * args.streams + 256
    return args.gpus > 1 and hard != resource.RLIM_INFINITY and need > hard


def build(args):
    """(engines, shards, front door or None): N = 1 listens directly (no extra hop), N > 1 goes through the front door."""
    from . import dist_util, engine, ingest
    blob, mode = load_blob(args)
    n = max(1, args.gpus)
    args.precision_plan = None
    choose_precision(args, mode)
    engines, shards, door = [], [], None
    taken = {}                                   # cores of a NUMA node's run already given to an earlier shard's front-end
    try:
        for r in range(n):
            dev = 0 if args.share_gpu else r
            eng = engine.Engine(blob, args.vap_process_rate, args.context_len_sec, max_streams=args.streams,
                                max_batch=min(args.streams, args.max_batch), mode=mode, device_id=dev,
                                groups=2,   # two intra-tick overlap groups: ragged ticks of a few hundred streams get 20 % shorter
                                split_f16=(args.precision == "split"))
            engines.append(eng)
            passive = n > 1
            cores = None
            if args.pin:                  # tick / receive / sender threads next to their GPU, every shard on cores of its own
                node_key = tuple(dist_util.gpu_node_cores(dev)[:1])
                nthr = 1 + args.rx_threads + args.tx_threads
                cores, _ = dist_util.front_end_placement(dev, nthr, skip=taken.get(node_key, 0))
                taken[node_key] = taken.get(node_key, 0) + nthr
            shards.append(ingest.NativeServer(eng, port_in=-1 if passive else args.port_num_in, port_out=-1 if passive else args.port_num_out,
                                              gain=args.audio_gain, max_wait_s=args.max_wait_ms * 1e-3, bind_any=args.bind_any,
                                              rx_threads=args.rx_threads, tx_threads=args.tx_threads, cores=cores))
        if n > 1:
            door = ingest.FrontDoor(shards, args.port_num_in, args.port_num_out, bind_any=args.bind_any)
    except Exception:
        teardown(engines, shards, door)
        raise
    return engines, shards, door, mode


def run_worker(args) -> int:
    """One GPU's worker process behind a front-door process: engine + passive front-end, connections arrive over the link (``--worker-link``)."""
    import os
    from . import dist_util, engine, ingest
    stop = {"now": False}
    signal.signal(signal.SIGTERM, lambda *_: stop.__setitem__("now", True))
    signal.signal(signal.SIGINT, signal.SIG_IGN)               # ^C goes to the whole foreground group: the door process stops us in order
    r = args.worker_rank
    try:
        blob, mode = load_blob(args)
        dev = 0 if args.share_gpu else r
        eng = engine.Engine(blob, args.vap_process_rate, args.context_len_sec, max_streams=args.streams, max_batch=min(args.streams, args.max_batch),
                            mode=mode, device_id=dev, groups=2, split_f16=(args.precision == "split"))
        cores = None
        if args.pin:
            nthr = 1 + args.rx_threads + args.tx_threads
            cores, _ = dist_util.front_end_placement(dev, nthr, skip=(r * nthr if args.share_gpu else 0))
        shard = ingest.NativeServer(eng, port_in=-1, port_out=-1, gain=args.audio_gain, max_wait_s=args.max_wait_ms * 1e-3,
                                    rx_threads=args.rx_threads, tx_threads=args.tx_threads, cores=cores)
        shard.attach_link(args.worker_link)
    except Exception as e:                                      # noqa: BLE001
        print(f"[vapx] GPU {r}: worker start-up failed: {e}", file=sys.stderr, flush=True)
        return 1
    parent = os.getppid()
    last = time.time()
    while not stop["now"] and os.getppid() == parent:          # an orphaned worker (the door process died) stops too
        time.sleep(0.2)
        if args.stats_sec > 0 and time.time() - last >= args.stats_sec:
            last = time.time()
            st = shard.stats(reset_latency_window=True)
            print(f"[vapx] GPU {r}: " + json.dumps({k: (round(v, 3) if isinstance(v, float) else v) for k, v in st.items()}), flush=True)
    shard.close()
    eng.close()
    return 0


def run_door(args, argv) -> int:
    """The front-door process of ``--worker-procs``: spawns one worker per GPU, owns the port pair, touches no GPU."""
    import subprocess
    from . import ingest
    stop = {"now": False}
    signal.signal(signal.SIGTERM, lambda *_: stop.__setitem__("now", True))
    signal.signal(signal.SIGINT, lambda *_: stop.__setitem__("now", True))
    n = max(1, args.gpus)
    mode = args.mode
    if args.precision == "auto":                                # plan once, here; the workers are told the result
        if mode is None and args.synthetic_weights is None:
            from . import checkpoints
            mode = checkpoints.infer_mode(checkpoints.load_state_dicts(args.vap_model, args.cpc_model)[1])
        try:
            choose_precision(args, mode or "vap")
        except Exception as e:                                  # noqa: BLE001
            print(f"[vapx] start-up failed: {e}", file=sys.stderr, flush=True)
            return 1
    base = [a for a in (argv if argv is not None else sys.argv[1:])]
    for flag in ("--precision", "--worker-procs"):              # the workers get the decided values
        while flag in base:
            k = base.index(flag)
            del base[k:k + 2]
    links, workers = [], []
    door = None
    try:
        for r in range(n):
            mine, theirs = ingest.link_pair()
            workers.append(subprocess.Popen([sys.executable, "-u", "-m", "vap_realtime_amd.serve"] + base + ["--precision", args.precision, "--worker-link",
                                             str(theirs.fileno()), "--worker-rank", str(r)], pass_fds=[theirs.fileno()]))
            theirs.close()
            links.append(mine)
        door = ingest.RemoteFrontDoor(links, args.port_num_in, args.port_num_out, bind_any=args.bind_any)
    except Exception as e:                                      # noqa: BLE001
        print(f"[vapx] start-up failed: {e}", file=sys.stderr, flush=True)
        for w in workers:
            w.terminate()
        return 1
    print(f"[vapx] {n} GPU(s) x {args.streams} dialogue slots in {n} worker processes, mode {mode or ('vap' if args.synthetic_weights is not None else 'from the state dict')}, {args.precision} arithmetic, "
          f"{args.vap_process_rate} Hz / {args.context_len_sec} s — input :{door.port_in}, output :{door.port_out} (front-door process: dialogue k -> GPU k mod N)", flush=True)
    rc = 0
    while not stop["now"]:
        time.sleep(0.2)
        dead = [r for r, w in enumerate(workers) if w.poll() is not None]
        if len(dead) == n:                                       # nobody left to serve
            print("[vapx] every worker has exited", file=sys.stderr, flush=True)
            rc = 1
            break
    door.close()
    for w in workers:
        if w.poll() is None:
            w.terminate()
    for w in workers:
        try:
            w.wait(timeout=30)
        except Exception:                                        # noqa: BLE001
            w.kill()
    for l in links:
        l.close()
    return rc


def teardown(engines, shards, door):
    if door is not None:
        door.close(close_shards=False)
    for s in shards:
        s.close()
    for e in engines:
        e.close()


def main(argv=None) -> int:
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
    ap.add_argument("--allow-overload", dest="allow_overload", action="store_true",
                    help="with --precision auto: start even when no arithmetic path holds <= 10 ms per frame at --streams dialogues per GPU")
    ap.add_argument("--precision", choices=["auto", "fp32", "split"], default="auto",
                    help="auto (default): fp32 while --streams x rate keeps the fp32 path <= 85 %% busy, else split, refusing loads neither path holds "
                         "within 10 ms (capacity.plan; the choice is printed).  fp32: every contraction on the fp32 MFMA (safe next to other tenants).  split: VAPX_FLAG_SPLIT_F16 — the same "
                         "contractions as fp32-accurate 3-term f16 split products, ~2x the streams per GPU at the same <= 1e-4 parity; for a "
                         "DEDICATED GPU/node (a process that issues f16 MFMAs all day can disturb co-running tenants: DESIGN.md \"co-running f16 MFMA\")")
    ap.add_argument("--share-gpu", dest="share_gpu", action="store_true", help="plumbing check on a 1-GPU box: every shard's engine on device 0")
    ap.add_argument("--max_wait_ms", type=float, default=2.0)
    ap.add_argument("--rx_threads", type=int, default=4)
    ap.add_argument("--tx_threads", type=int, default=4)
    ap.add_argument("--bind_any", action="store_true", help="listen on 0.0.0.0 instead of 127.0.0.1")
    ap.add_argument("--pin", dest="pin", action="store_true",
                    help="pin every shard's tick / receive / sender threads to consecutive cores at the top of its GPU's NUMA node (default: the "
                         "scheduler places them).  Worth it on a host you own — keep everything else off those cores; on a shared host it measured "
                         "no better than floating threads (profiles/r05_frontend/README.md)")
    ap.add_argument("--stats_sec", type=float, default=10.0)
    ap.add_argument("--synthetic-weights", dest="synthetic_weights", type=int, default=None)
    ap.add_argument("--worker-procs", dest="worker_procs", choices=["auto", "on", "off"], default="auto",
                    help="one worker PROCESS per GPU behind a front-door process that passes accepted connections on (auto: when one process could not "
                         "hold 2 sockets per dialogue under RLIMIT_NOFILE; --gpus 8 x --streams 4096 needs 65 792 descriptors)")
    ap.add_argument("--worker-link", dest="worker_link", type=int, default=None, help=argparse.SUPPRESS)     # set by the door process
    ap.add_argument("--worker-rank", dest="worker_rank", type=int, default=0, help=argparse.SUPPRESS)
    args = ap.parse_args(argv)
    if args.worker_link is not None:
        return run_worker(args)
    if wants_worker_procs(args):
        return run_door(args, argv)
    stop = {"now": False}
    signal.signal(signal.SIGTERM, lambda *_: stop.__setitem__("now", True))   # the normal service-stop signal: shut every GPU down in order
    signal.signal(signal.SIGINT, lambda *_: stop.__setitem__("now", True))
    try:
        engines, shards, door, mode = build(args)
    except Exception as e:                                      # noqa: BLE001
        print(f"[vapx] start-up failed: {e}", file=sys.stderr, flush=True)
        return 1
    pin, pout = (door.port_in, door.port_out) if door else (shards[0].port_in, shards[0].port_out)
    print(f"[vapx] {len(engines)} GPU(s) x {args.streams} dialogue slots, mode {mode}, {args.precision} arithmetic, {args.vap_process_rate} Hz / {args.context_len_sec} s — "
          f"input :{pin}, output :{pout}" + (" (front door: dialogue k -> GPU k mod N)" if door else ""), flush=True)
    last = time.time()
    while not stop["now"]:
        time.sleep(0.2)
        if args.stats_sec > 0 and time.time() - last >= args.stats_sec:
            last = time.time()
            for r, s in enumerate(shards):
                st = s.stats(reset_latency_window=True)
                print(f"[vapx] GPU {r}: " + json.dumps({k: (round(v, 3) if isinstance(v, float) else v) for k, v in st.items()}), flush=True)
    teardown(engines, shards, door)
    return 0


if __name__ == "__main__":
    sys.exit(main())
