#!/usr/bin/env python3
"""End-to-end capacity of the TCP front-ends with the reference's wire format, driven by the native real-time load
generator (tools/loadgen: S dialogue clients sending 10 ms packets in real time, frame boundaries spread over the frame
period; latency = last byte of a frame sent -> complete result packet received).

  tools/server_load.py --streams 4096 --seconds 20                 native front-end (libvapx vapx_ingest_*) + engine on the GPU
  tools/server_load.py --streams 1024 --python                     the Python twin (server.ManyStreamServer)
  tools/server_load.py --streams 256 --fake                        native front-end over a trivial step function (no GPU)

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
    ap.add_argument("--no-pin", action="store_true", help="do not place the front-end's threads / the load generator on disjoint cores")
    ap.add_argument("--pin-mode", default="clients-only", choices=["each", "set", "clients-only"],
                    help="each: one core per front-end thread; set: the front-end's threads share the core range as one affinity set (16 cores); "
                         "clients-only: the front-end floats, only the load generator is kept off the GPU node's top cores")
    args = ap.parse_args()
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
    if args.fake:
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
    cmd = [loadgen, "--port-in", str(srv.port_in), "--port-out", str(srv.port_out), "--streams", str(S), "--hz", str(args.hz),
           "--seconds", str(args.seconds), "--warm", str(args.warm), "--packet-ms", str(args.packet_ms), "--threads", str(args.client_threads)]
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
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, preexec_fn=client_affinity)
    base = {}
    if hasattr(srv, "stats"):                       # server-side latency window = the load generator's measured window
        time.sleep(args.warm + 1.0)
        base = srv.stats(reset_latency_window=True)
    out = proc.communicate(timeout=args.seconds + args.warm + 120)[0].decode()
    res = json.loads(out.strip().splitlines()[-1])
    net1 = net_counters()
    res["net_counters_delta"] = {k: (net1[k] - net0.get(k, 0) if k != "netdev_max_backlog" else net1[k]) for k in net1}
    res["host_limits"] = host_limits()
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
