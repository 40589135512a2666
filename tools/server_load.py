#!/usr/bin/env python3
"""End-to-end capacity of the Python TCP front-end (server.ManyStreamServer + libvapx) with the reference's wire format:
K client processes drive S dialogue streams closed-loop (send one frame = hop/160 packets of 2560 B, wait for the 12.9 KB
result packet, repeat) as fast as the server answers.  Prints stream-frames/s and the real-time stream count that equals.
Every client runs the same fixed number of rounds: the server pairs the k-th output connection with the k-th stream, so a
client may be listening to another client's streams and all of them have to send the same number of frames.
Usage: tools/server_load.py [streams] [client procs] [rounds]"""
import multiprocessing as mp
import socket
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])


def client(args):
    port_in, port_out, n, hop, rounds, seed = args
    from vap_realtime_amd import synth, wire
    audio = synth.dialogue_batch([seed], hop * 8)[0].astype(np.float64)          # [2, hop*8]
    frames = [wire.encode_input(audio[0, f * hop:(f + 1) * hop], audio[1, f * hop:(f + 1) * hop]) for f in range(8)]
    ins, outs = [], []
    for _ in range(n):
        s = socket.create_connection(("127.0.0.1", port_in)); s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1); ins.append(s)
    time.sleep(0.5)
    for _ in range(n):
        s = socket.create_connection(("127.0.0.1", port_out)); outs.append(s)
    time.sleep(1.0)
    done, t0 = 0, time.time()
    for f in range(rounds):
        for s in ins:
            s.sendall(frames[f % 8])
        for s in outs:
            hdr = b""
            while len(hdr) < 4:
                hdr += s.recv(4 - len(hdr))
            need = int.from_bytes(hdr, "little")
            while need > 0:
                need -= len(s.recv(min(need, 1 << 16)))
        done += n
    return done, time.time() - t0


def main():
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    K = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 300
    from vap_realtime_amd import realtime, weights as W
    from vap_realtime_amd.server import ManyStreamServer
    cpc, vap_sd = W.synthetic_weights(0, 20)
    vap = realtime.ManyStreamVAP(cpc, vap_sd, 20, 2.5, n_streams=S)
    srv = ManyStreamServer(vap, port_in=0, port_out=0, max_wait_s=0.002).start()
    per = S // K
    with mp.get_context("spawn").Pool(K) as pool:
        res = pool.map(client, [(srv.port_in, srv.port_out, per, 800, rounds, 100 + i) for i in range(K)])
    srv.stop()
    frames = sum(r[0] for r in res)
    dt = max(r[1] for r in res)
    print(f"{per * K} streams over {K} client processes: {frames / dt:.0f} stream-frames/s end to end "
          f"(= {frames / dt / 20:.0f} real-time 20 Hz streams through one Python server process)")


if __name__ == "__main__":
    main()
