#!/usr/bin/env python3
"""End-to-end capacity of the TCP front-ends with the reference's wire format, driven by the native real-time load
generator (tools/loadgen: S dialogue clients sending 10 ms packets in real time, frame boundaries spread over the frame
period; latency = last byte of a frame sent -> complete result packet received).

  tools/server_load.py --streams 4096 --seconds 20                 native front-end (libvapx vapx_ingest_*) + engine on the GPU
  tools/server_load.py --streams 1024 --python                     the Python twin (server.ManyStreamServer)
  tools/server_load.py --streams 256 --fake                        native front-end over a trivial step function (no GPU)
  tools/server_load.py --standin --shards 8 --streams 32768 --loadgen-procs 4
                                                                   the HOST half of BASELINE config 4: ONE front door over 8 passive front-ends whose
                                                                   step function is a native stand-in for the GPU tick (tools/standin_step.cpp: touches
                                                                   the audio, writes result rows, sleeps the measured tick time), several load-generator
                                                                   processes (a process here may hold ~20 k descriptors), latency stamps in-band

Prints one JSON line: the load generator's view (frames answered, latency percentiles) + the server's own counters.
"""
import argparse
import json
import os
import subprocess
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def net_counters():
    """Kernel-side evidence for client-visible stalls: TCP retransmissions / timeouts (/proc/net/snmp, /proc/net/netstat) and packets the
    per-CPU softnet backlog DROPPED (/proc/net/softnet_stat, 2nd column; the loopback device hands every segment to that queue, bounded by
    net.core.netdev_max_backlog).  A dropped loopback segment comes back only after the retransmission timeout (>= 200 ms)."""
    out = {}
    try:
        with open("/proc/net/snmp") as f:
            rows = [l.split() for l in f if l.startswith("Tcp:")]
        out.update({k: int(v) for k, v in zip(rows[0][1:], rows[1][1:]) if k in ("RetransSegs", "OutSegs", "InSegs")})
    except Exception:
        pass
    try:
        with open("/proc/net/netstat") as f:
            rows = [l.split() for l in f if l.startswith("TcpExt:")]
        keep = ("TCPTimeouts", "TCPLossProbes", "TCPLostRetransmit", "TCPBacklogDrop", "TCPRcvQDrop", "ListenDrops", "ListenOverflows", "TCPSynRetrans",
                "TCPFastRetrans", "TCPSlowStartRetrans", "PruneCalled", "TCPRcvCollapsed", "TCPDelivered", "TCPSackRecovery")
        out.update({k: int(v) for k, v in zip(rows[0][1:], rows[1][1:]) if k in keep})
    except Exception:
        pass
    try:
        with open("/proc/net/softnet_stat") as f:
            cols = [[int(x, 16) for x in l.split()] for l in f]
        out["softnet_dropped"] = sum(c[1] for c in cols)
        out["softnet_time_squeeze"] = sum(c[2] for c in cols)
    except Exception:
        pass
    try:
        with open("/proc/sys/net/core/netdev_max_backlog") as f:
            out["netdev_max_backlog"] = int(f.read())
    except Exception:
        pass
    # CPU-bandwidth throttling of this container (cgroup v2 cpu.stat / v1 cpu.stat): a cgroup that exhausts its quota is frozen until the
    # next 100 ms period — every thread of server AND load generator stalls at once, whatever their placement
    for path in ("/sys/fs/cgroup/cpu.stat", "/sys/fs/cgroup/cpu/cpu.stat", "/sys/fs/cgroup/cpu,cpuacct/cpu.stat"):
        try:
            with open(path) as f:
                for l in f:
                    k, v = l.split()
                    if k in ("nr_periods", "nr_throttled", "throttled_usec", "throttled_time", "usage_usec"):
                        out["cgroup_" + k] = int(v)
            break
        except Exception:
            continue
    return out


def host_limits():
    out = {"cpus_allowed": len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else os.cpu_count()}
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                out["cgroup_cpu_max"] = f.read().strip()
            break
        except Exception:
            continue
    try:
        with open("/proc/loadavg") as f:
            out["loadavg"] = f.read().split()[:3]
    except Exception:
        pass
    return out


def standin_lib():
    import ctypes as C
    so = os.path.join(ROOT, "tools", "libstandin_step.so")
    if not os.path.exists(so):
        subprocess.check_call(["g++", "-O2", "-shared", "-fPIC", "-o", so, os.path.join(ROOT, "tools", "standin_step.cpp")])
    lib = C.CDLL(so)

    class Cfg(C.Structure):
        _fields_ = [("hop", C.c_int32), ("out_stride", C.c_int32), ("base_us", C.c_int32), ("per_stream_ns", C.c_int32), ("ticks", C.c_int64), ("frames", C.c_int64)]
    return lib, Cfg, C.cast(lib.standin_step, C.c_void_p).value


def standin_worker(args):
    """One stand-in shard in a process of its own (--worker-procs): passive front-end over the native stand-in step, linked to the parent's front
    door.  Answers the parent's requests on stdin: "stats" / "reset" (reset the latency window) -> one JSON line; EOF or "quit" -> exit."""
    import ctypes as C
    import signal
    from vap_realtime_amd import engine, ingest
    signal.signal(signal.SIGINT, signal.SIG_IGN)
    lib, Cfg, step_ptr = standin_lib()
    cfg = Cfg(16000 // args.hz, engine.OUT_STRIDE, args.standin_base_us, args.standin_per_stream_ns, 0, 0)
    per = args.streams
    sh = ingest.NativeServer.over_native_function(step_ptr, C.addressof(cfg), per, args.hz, max_batch=args.max_batch or per, keep=(lib, cfg), port_in=-1, port_out=-1,
                                                  max_wait_s=args.max_wait_ms * 1e-3, min_batch=args.min_batch, rx_threads=args.rx_threads,
                                                  tx_threads=args.tx_threads, target_util=args.target_util)
    sh.attach_link(args.standin_worker_link)
    t0 = os.times()
    for line in sys.stdin:
        cmd = line.strip()
        if cmd == "quit":
            break
        st = sh.stats(reset_latency_window=(cmd == "reset"))
        t1 = os.times()
        st["cpu_s"] = (t1.user + t1.system) - (t0.user + t0.system)
        print(json.dumps(st), flush=True)
    sh.close()
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=256)
    ap.add_argument("--hz", type=int, default=20)
    ap.add_argument("--ctx-sec", type=float, default=2.5)
    ap.add_argument("--seconds", type=float, default=10.0)
    ap.add_argument("--warm", type=float, default=4.0)
    ap.add_argument("--packet-ms", type=int, default=10)
    ap.add_argument("--max-wait-ms", type=float, default=2.0)
    ap.add_argument("--min-batch", type=int, default=0)
    ap.add_argument("--max-batch", type=int, default=0)
    ap.add_argument("--target-util", type=float, default=0.9)
    ap.add_argument("--rx-threads", type=int, default=4)
    ap.add_argument("--tx-threads", type=int, default=4)
    ap.add_argument("--client-threads", type=int, default=8)
    ap.add_argument("--groups", type=int, default=2, help="intra-tick overlap groups of the engine (what vap_realtime_amd.serve uses: 2)")
    ap.add_argument("--repeat", type=int, default=1, help="run the load generator this many times against the SAME server (slot reuse, resets at scale)")
    ap.add_argument("--python", action="store_true", help="serve with the Python front-end instead of the native one")
    ap.add_argument("--shards", type=int, default=1, help="N engines (all on this GPU: a one-GPU box) behind ONE front door (vapx_frontdoor_*), "
                    "--streams / N slots each: what `python -m vap_realtime_amd.serve --gpus N --share-gpu` runs")
    ap.add_argument("--devices", default="", help="with --shards N: comma list of N device ids, one engine per listed GPU (default: all on device 0)")
    ap.add_argument("--split-f16", action="store_true", help="engines on the opt-in split-precision path (serve --precision split)")
    ap.add_argument("--mode", default="vap", choices=["vap", "bc", "nod"])
    ap.add_argument("--backlog", type=int, default=0, help="try to raise net.core.netdev_max_backlog to this before the run (needs root; 0 = leave it)")
    ap.add_argument("--fake", action="store_true", help="native front-end over a trivial step function (plumbing only, no GPU)")
    ap.add_argument("--standin", action="store_true", help="native front-end(s) over the native stand-in for the GPU tick (no GPU, no Python on the tick thread); "
                    "with --shards N: N passive front-ends behind ONE front door")
    ap.add_argument("--standin-base-us", type=int, default=800, help="stand-in tick time = base + per-stream x n (measured with the real engine: 1.72 ms at n = 165)")
    ap.add_argument("--standin-per-stream-ns", type=int, default=5600)
    ap.add_argument("--worker-procs", action="store_true", help="with --standin --shards N: every shard in a PROCESS of its own behind a front-door process "
                    "(vapx_frontdoor_open_links: accepted sockets are passed on) - what `serve --worker-procs on` runs; needed beyond ~9 800 dialogues under a "
                    "20 000-descriptor limit")
    ap.add_argument("--standin-worker-link", type=int, default=None, help=argparse.SUPPRESS)     # set by the parent: this process IS a stand-in worker
    ap.add_argument("--loadgen-procs", type=int, default=1, help="load-generator processes (dialogues dealt round-robin; needs in-band latency stamps)")
    ap.add_argument("--src-ips", type=int, default=0, help="load generators dial from this many 127.0.0.x source addresses (0: the default address)")
    ap.add_argument("--inband", action="store_true", help="latency stamps travel in the audio and come back in the echoed result packet (implied by --loadgen-procs > 1)")
    ap.add_argument("--no-pin", action="store_true", help="do not place the front-end's threads / the load generator on disjoint cores")
    ap.add_argument("--pin-mode", default="clients-only", choices=["each", "set", "clients-only"],
                    help="each: one core per front-end thread; set: the front-end's threads share the core range as one affinity set (16 cores); "
                         "clients-only: the front-end floats, only the load generator is kept off the GPU node's top cores")
    args = ap.parse_args()
    if args.standin_worker_link is not None:
        return standin_worker(args)
    loadgen = os.path.join(ROOT, "tools", "loadgen")
    if not os.path.exists(loadgen):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "vap-realtime_amd", "csrc"), "../../tools/loadgen"])
    S = args.streams
    from vap_realtime_amd import dist_util, engine, ingest
    srv = None
    # placement: the front-end's tick / receive / sender threads on consecutive cores of the GPU's NUMA node, the load generator on the
    # other cores (another node's first) — 4096 client sockets' softirq work otherwise lands on whatever core a front-end thread runs on
    # (round 4, driver's box: client-side p99 11.97 ms against 3.9-5.2 on the builder's boxes)
    cores, client_cores = (None, None)
    if not args.no_pin and not args.python:
        first_dev = int(args.devices.split(",")[0]) if args.devices else 0
        cores, client_cores = dist_util.front_end_placement(first_dev, 16 if args.pin_mode != "each" else 1 + args.rx_threads + args.tx_threads)
        if args.pin_mode == "clients-only":
            cores = None
    if args.standin and args.worker_procs:
        N = max(1, args.shards)
        per = (S + N - 1) // N
        links, workers = [], []
        for k in range(N):
            mine, theirs = ingest.link_pair()
            workers.append(subprocess.Popen([sys.executable, "-u", os.path.abspath(__file__), "--standin-worker-link", str(theirs.fileno()), "--streams", str(per),
                                             "--hz", str(args.hz), "--max-wait-ms", str(args.max_wait_ms), "--min-batch", str(args.min_batch), "--max-batch", str(args.max_batch),
                                             "--rx-threads", str(args.rx_threads), "--tx-threads", str(args.tx_threads), "--target-util", str(args.target_util),
                                             "--standin-base-us", str(args.standin_base_us), "--standin-per-stream-ns", str(args.standin_per_stream_ns)],
                                            pass_fds=[theirs.fileno()], stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True))
            theirs.close()
            links.append(mine)
        door = ingest.RemoteFrontDoor(links, 0, 0)

        def ask(cmd):
            for w in workers:
                w.stdin.write(cmd + "\n")
                w.stdin.flush()
            return [json.loads(w.stdout.readline()) for w in workers]

        class _Procs:
            port_in, port_out = door.port_in, door.port_out

            @staticmethod
            def stats(reset_latency_window=False):
                per_shard = ask("reset" if reset_latency_window else "stats")
                tot = {k: sum(p_[k] for p_ in per_shard) for k in ("frames_done", "ticks", "overruns", "dropped_listeners", "numeric_resets", "late_over_10ms", "answered",
                                                                   "rx_bytes", "tx_bytes", "in_connections", "out_connections", "cpu_s")}
                tot["lat_p50_ms"] = float(np.median([p_["lat_p50_ms"] for p_ in per_shard]))
                tot["lat_p99_ms"] = max(p_["lat_p99_ms"] for p_ in per_shard)
                tot["lat_max_ms"] = max(p_["lat_max_ms"] for p_ in per_shard)
                tot["mean_batch"] = float(np.mean([p_["mean_batch"] for p_ in per_shard]))
                tot["step_mean_ms"] = float(np.mean([p_["step_mean_ms"] for p_ in per_shard]))
                tot["per_shard"] = per_shard
                tot["front_door"] = door.counts()
                return tot

            @staticmethod
            def stop():
                door.close()
                for w in workers:
                    try:
                        w.stdin.write("quit\n")
                        w.stdin.flush()
                        w.stdin.close()
                    except Exception:                 # noqa: BLE001
                        pass
                for w in workers:
                    try:
                        w.wait(timeout=30)
                    except Exception:                 # noqa: BLE001
                        w.kill()
                for l in links:
                    l.close()
        srv = _Procs
        cores = client_cores = None
        kind = (f"front-door PROCESS (vapx_frontdoor_open_links) + {N} worker processes, each a passive native front-end over the native stand-in step "
                f"function ({args.standin_base_us} us + {args.standin_per_stream_ns} ns x n per tick), {per} dialogue slots each")
    elif args.standin:
        import ctypes as C
        lib, Cfg, step_ptr = standin_lib()
        N = max(1, args.shards)
        per = (S + N - 1) // N
        cfgs = [Cfg(16000 // args.hz, engine.OUT_STRIDE, args.standin_base_us, args.standin_per_stream_ns, 0, 0) for _ in range(N)]
        nthr = 1 + args.rx_threads + args.tx_threads
        passive = N > 1
        shards = [ingest.NativeServer.over_native_function(step_ptr, C.addressof(cfgs[k]), per, args.hz, max_batch=args.max_batch or per, keep=(lib, cfgs),
                                                           port_in=-1 if passive else 0, port_out=-1 if passive else 0, max_wait_s=args.max_wait_ms * 1e-3,
                                                           min_batch=args.min_batch, rx_threads=args.rx_threads, tx_threads=args.tx_threads, target_util=args.target_util)
                  for k in range(N)]
        door = ingest.FrontDoor(shards, 0, 0) if passive else None

        class _Standin:
            port_in, port_out = (door.port_in, door.port_out) if door else (shards[0].port_in, shards[0].port_out)

            @staticmethod
            def stats(reset_latency_window=False):
                per_shard = [sh.stats(reset_latency_window) for sh in shards]
                tot = {k: sum(p_[k] for p_ in per_shard) for k in ("frames_done", "ticks", "overruns", "dropped_listeners", "numeric_resets", "late_over_10ms", "answered",
                                                                   "rx_bytes", "tx_bytes", "in_connections", "out_connections")}
                tot["lat_p50_ms"] = float(np.median([p_["lat_p50_ms"] for p_ in per_shard]))
                tot["lat_p99_ms"] = max(p_["lat_p99_ms"] for p_ in per_shard)          # the worst shard's window
                tot["lat_max_ms"] = max(p_["lat_max_ms"] for p_ in per_shard)
                tot["mean_batch"] = float(np.mean([p_["mean_batch"] for p_ in per_shard]))
                tot["step_mean_ms"] = float(np.mean([p_["step_mean_ms"] for p_ in per_shard]))
                tot["per_shard"] = per_shard
                if door:
                    tot["front_door"] = door.counts()
                return tot

            @staticmethod
            def stop():
                if door:
                    door.close()
                else:
                    shards[0].close()
        srv = _Standin
        cores = client_cores = None
        kind = (f"ONE front door (vapx_frontdoor_*) + {N} passive native front-ends over the native stand-in step function "
                f"({args.standin_base_us} us + {args.standin_per_stream_ns} ns x n per tick), {per} dialogue slots each" if passive else
                "native front-end over the native stand-in step function")
    elif args.fake:
        def step(ids, audio, out):
            out[:, 0:2] = np.abs(audio).mean(axis=2)
            return 0
        srv = ingest.NativeServer.over_function(step, S, args.hz, max_batch=args.max_batch or S, max_wait_s=args.max_wait_ms * 1e-3,
                                                min_batch=args.min_batch, rx_threads=args.rx_threads, tx_threads=args.tx_threads)
        cores = client_cores = None                       # (over_function has no placement argument)
        kind = "native front-end over a trivial step function"
    else:
        from vap_realtime_amd import realtime, weights as W
        cpc, vap_sd = W.synthetic_weights(0, args.hz, args.mode)
        if args.python:
            from vap_realtime_amd.server import ManyStreamServer
            vap = realtime.ManyStreamVAP(cpc, vap_sd, args.hz, args.ctx_sec, n_streams=S, max_batch=args.max_batch or None)
            srv = ManyStreamServer(vap, port_in=0, port_out=0, max_wait_s=args.max_wait_ms * 1e-3).start()
            kind = "Python front-end (server.ManyStreamServer)"
        elif args.shards > 1:
            N = args.shards
            blob = W.pack_blob(cpc, vap_sd, args.mode)
            devs = [int(d) for d in args.devices.split(",") if d != ""] or [0] * N
            assert len(devs) == N, "--devices needs one id per shard"
            engs = [engine.Engine(blob, args.hz, args.ctx_sec, max_streams=(S + N - 1) // N, max_batch=args.max_batch or None, groups=args.groups,
                                  device_id=devs[k], mode=args.mode, split_f16=args.split_f16) for k in range(N)]
            shards = [ingest.NativeServer(e, port_in=-1, port_out=-1, max_wait_s=args.max_wait_ms * 1e-3, min_batch=args.min_batch,
                                          rx_threads=args.rx_threads, tx_threads=args.tx_threads, target_util=args.target_util,
                                          cores=(None if args.no_pin else dist_util.front_end_placement(devs[k], 1 + args.rx_threads + args.tx_threads,
                                                                                                     skip=k * (1 + args.rx_threads + args.tx_threads))[0]))
                      for k, e in enumerate(engs)]
            door = ingest.FrontDoor(shards, 0, 0)

            class _Srv:                                   # the load generator's view: one port pair; statistics summed over the shards
                port_in, port_out = door.port_in, door.port_out

                @staticmethod
                def stats(reset_latency_window=False):
                    per = door.stats(reset_latency_window)
                    tot = {k: sum(p[k] for p in per) for k in ("frames_done", "ticks", "overruns", "dropped_listeners", "numeric_resets")}
                    tot["per_shard"] = per
                    tot["front_door"] = door.counts()
                    return tot

                @staticmethod
                def stop():
                    door.close()
                    for e in engs:
                        e.close()
            srv = _Srv
            kind = f"ONE front door (vapx_frontdoor_*) + {N} passive native front-ends + {N} engines on devices {devs}"
        else:
            eng = engine.Engine(W.pack_blob(cpc, vap_sd, args.mode), args.hz, args.ctx_sec, max_streams=S, max_batch=args.max_batch or None, groups=args.groups,
                                mode=args.mode, split_f16=args.split_f16)
            srv = ingest.NativeServer(eng, port_in=0, port_out=0, max_wait_s=args.max_wait_ms * 1e-3, min_batch=args.min_batch,
                                      rx_threads=args.rx_threads, tx_threads=args.tx_threads, target_util=args.target_util, cores=cores,
                                      core_set=(args.pin_mode == "set"))
            kind = "native front-end (vapx_ingest_*) + engine"
    P = max(1, args.loadgen_procs)
    inband = args.inband or P > 1
    per_proc = [len(range(r, S, P)) for r in range(P)]
    import tempfile
    sync_dir = tempfile.mkdtemp(prefix="loadgen_sync_")

    def cmd_for(r):
        c = [loadgen, "--port-in", str(srv.port_in), "--port-out", str(srv.port_out), "--streams", str(per_proc[r]), "--hz", str(args.hz),
             "--seconds", str(args.seconds), "--warm", str(args.warm), "--packet-ms", str(args.packet_ms), "--threads", str(args.client_threads)]
        if inband:
            c += ["--inband", "1", "--hist-out", os.path.join(sync_dir, f"hist.{r}.json")]
        if P > 1:
            c += ["--procs", str(P), "--rank", str(r), "--total-streams", str(S), "--sync-dir", sync_dir]
        if args.src_ips:
            c += ["--src-ips", str(args.src_ips)]
        return c
    cmd = cmd_for(0)

    def client_affinity():                          # the load generator (and its threads) stays off the front-end's cores
        if client_cores:
            try:
                os.sched_setaffinity(0, client_cores)
            except Exception:                         # noqa: BLE001
                pass
    for rep in range(args.repeat - 1):              # earlier rounds: only their summary line is kept
        first = json.loads(subprocess.run(cmd, stdout=subprocess.PIPE, preexec_fn=client_affinity).stdout.decode().strip().splitlines()[-1])
        print(json.dumps({"round": rep, "frames_answered": first["frames_answered"], "frames_sent": first["frames_sent"],
                          "lat_p99_ms": first["lat_p99_ms"]}), file=sys.stderr)
        time.sleep(1.0)
    if args.backlog > 0:
        try:
            with open("/proc/sys/net/core/netdev_max_backlog", "w") as f:
                f.write(str(args.backlog))
        except Exception as e:                                    # noqa: BLE001
            print(f"could not set netdev_max_backlog: {e}", file=sys.stderr)
    net0 = net_counters()
    cpu0 = (time.time(), os.times())
    procs = [subprocess.Popen(cmd_for(r), stdout=subprocess.PIPE, preexec_fn=client_affinity) for r in range(P)]
    base = {}
    if hasattr(srv, "stats"):                       # server-side latency window = the load generator's measured window
        time.sleep(args.warm + 1.0 + (2.5 + 2e-4 * S if P > 1 else 0.0))
        base = srv.stats(reset_latency_window=True)
        cpu_w0 = (time.time(), os.times())
    outs = [p_.communicate(timeout=args.seconds + args.warm + 300)[0].decode() for p_ in procs]
    cpu1 = (time.time(), os.times())
    parts = [json.loads(o.strip().splitlines()[-1]) for o in outs]
    res = parts[0]
    if P > 1:                                       # merge the processes: counters add up, percentiles from the summed histograms
        for k in ("frames_sent", "frames_answered", "unanswered_at_end", "latency_samples", "schedule_slips", "route_changes", "inband_unreadable", "streams"):
            res[k] = sum(p_[k] for p_ in parts)
        late_key = [k for k in res if k.startswith("late_over_")][0]
        res[late_key] = sum(p_[late_key] for p_ in parts)
        for k in ("client_max_send_lag_ms", "client_max_send_call_ms", "client_max_recv_pass_ms", "lat_max_ms"):
            res[k] = max(p_[k] for p_ in parts)
        res["stream_frames_per_s"] = sum(p_["stream_frames_per_s"] for p_ in parts)
        res["stall_events_over_50ms"] = sum((p_["stall_events_over_50ms"] for p_ in parts), [])[:24]
        res["streams_with_unanswered_frames"] = sum((p_["streams_with_unanswered_frames"] for p_ in parts), [])[:24]
        res["per_process"] = [{k: p_[k] for k in ("rank", "streams", "frames_sent", "frames_answered", "lat_p50_ms", "lat_p99_ms", "lat_max_ms", "schedule_slips")} for p_ in parts]
    if inband:
        hist = None
        for r in range(P):
            h = json.load(open(os.path.join(sync_dir, f"hist.{r}.json")))
            hist = np.asarray(h["hist"], dtype=np.int64) if hist is None else hist + np.asarray(h["hist"], dtype=np.int64)
        cum = np.cumsum(hist)
        if cum[-1] > 0:
            for q, name in ((0.5, "lat_p50_ms"), (0.99, "lat_p99_ms"), (0.999, "lat_p999_ms")):
                res[name] = float((np.searchsorted(cum, q * cum[-1]) + 1) * 0.05)      # upper edge of the bin (50 us resolution)
        res["procs"] = P
    import shutil
    shutil.rmtree(sync_dir, ignore_errors=True)
    net1 = net_counters()
    res["net_counters_delta"] = {k: (net1[k] - net0.get(k, 0) if k != "netdev_max_backlog" else net1[k]) for k in net1}
    res["host_limits"] = host_limits()
    # CPU the whole test burned: this process (the server's threads) and its children (the load generators), in cores
    wall = cpu1[0] - cpu0[0]
    res["cpu_cores_used"] = {"server_process": round(((cpu1[1].user + cpu1[1].system) - (cpu0[1].user + cpu0[1].system)) / wall, 2),
                             "load_generators": round(((cpu1[1].children_user + cpu1[1].children_system) - (cpu0[1].children_user + cpu0[1].children_system)) / wall, 2),
                             "wall_s": round(wall, 2)}
    res["server"] = kind
    res["placement"] = {"front_end_cores": list(cores) if cores else None, "client_cores": ([client_cores[0], client_cores[-1], len(client_cores)] if client_cores else None)}
    if hasattr(srv, "stats"):
        st = srv.stats()
        res["server_stats"] = st
        res["server_window"] = {k: st[k] - base.get(k, 0) for k in ("frames_done", "ticks", "overruns", "dropped_listeners", "numeric_resets")}
    res["realtime_streams_served"] = res["stream_frames_per_s"] / args.hz
    srv.stop()
    print(json.dumps(res))


if __name__ == "__main__":
    main()
