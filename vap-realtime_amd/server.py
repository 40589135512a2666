"""Many-stream TCP front-end that keeps the reference's packet framing.

The reference serves ONE dialogue per process: ``proc_serv_in`` listens with backlog 1, accepts a
single client and runs inference inline in the receive loop (rvap/vap_main/vap_main.py:354-414);
``proc_serv_out``/``proc_serv_out_dist`` broadcast every result to all output sockets (:338-352,
:416-457).  Here one process drives S dialogues on one GPU:

* every connection accepted on ``port_in`` becomes a stream (lowest free slot); it sends the same
  2560-byte packets (160 x {f64 ch1, f64 ch2});
* every connection accepted on ``port_out`` is attached to a stream: by default the k-th output
  connection listens to the k-th input stream; with ``broadcast=True`` (or a 1-stream server)
  every output receives every result, which is exactly the reference's behaviour;
* a tick runs when every connected stream has a complete frame, or ``max_wait_s`` after the first
  one became ready (ragged ticks: only the ready streams are stepped);
* result packets are byte-identical in layout to ``util.conv_vapresult_2_bytearray``.

The model object only needs ``.hop``, ``.mode``, ``.n_streams``, ``.reset(id)`` and
``.process(new_samples, ids, on_numeric="status")`` (``realtime.ManyStreamVAP``): with ``on_numeric="status"`` a non-finite stream
is reported through ``result["status"]`` instead of an exception, so that the other streams of the tick are still served.  A model
whose ``process`` only takes ``(new_samples, ids)`` works too (no ``status`` key = nothing flagged).
"""
from __future__ import annotations

import inspect
import selectors
import socket
import threading
import time
from typing import Dict, List, Optional

import numpy as np

from . import wire


class ManyStreamServer:
    def __init__(self, vap, port_in: int = 50007, port_out: int = 50008, host: str = "127.0.0.1", gain: float = 1.0,
                 max_wait_s: float = 0.004, broadcast: Optional[bool] = None, reset_on_connect: bool = True):
        self.vap = vap
        # which contract the model's process() has is decided ONCE, from its signature (a blanket `except TypeError` around the call would
        # also swallow a TypeError raised INSIDE a model that does take on_numeric, and step its state a second time: advisor r04)
        try:
            params = inspect.signature(vap.process).parameters
            self._status_contract = "on_numeric" in params or any(p.kind is inspect.Parameter.VAR_KEYWORD for p in params.values())
        except (TypeError, ValueError):
            self._status_contract = False
        self.S = vap.n_streams
        self.hop = vap.hop
        self.mode = getattr(vap, "mode", "vap")
        self.broadcast = (self.S == 1) if broadcast is None else broadcast
        self.reset_on_connect = reset_on_connect
        self.max_wait_s = max_wait_s
        self.asm = wire.PacketAssembler(self.S, self.hop, gain)
        self.sel = selectors.DefaultSelector()
        self.lin = self._listen(host, port_in)
        self.lout = self._listen(host, port_out)
        self.port_in = self.lin.getsockname()[1]
        self.port_out = self.lout.getsockname()[1]
        self.sel.register(self.lin, selectors.EVENT_READ, ("accept_in", None))
        self.sel.register(self.lout, selectors.EVENT_READ, ("accept_out", None))
        self.in_conn: Dict[int, socket.socket] = {}      # stream id -> input socket
        self.rxbuf: Dict[int, bytearray] = {}
        self.out_conns: List[List[socket.socket]] = [[] for _ in range(self.S)]
        self.out_all: List[socket.socket] = []
        self.first_ready_t: Optional[float] = None
        self.frames_done = 0
        self.numeric_resets = 0          # streams reset because their results went non-finite
        self._stop = threading.Event()
        self._thread: Optional[threading.Thread] = None

    @staticmethod
    def _listen(host, port):
        s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
        s.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        s.bind((host, port))
        s.listen(4096)     # a whole shard of dialogue clients may connect at once (capped by net.core.somaxconn)
        s.setblocking(False)
        return s

    # ---- connection management ----------------------------------------------------------------
    def _accept_in(self):
        conn, _ = self.lin.accept()
        free = [i for i in range(self.S) if i not in self.in_conn]
        if not free:
            conn.close()
            return
        sid = free[0]
        conn.setblocking(False)
        self.in_conn[sid] = conn
        self.rxbuf[sid] = bytearray()
        self.asm.fill[sid] = 0
        if self.reset_on_connect:
            self.vap.reset(sid)       # a new dialogue starts from a clean context (the reference keeps stale state)
        self.sel.register(conn, selectors.EVENT_READ, ("data", sid))

    def _accept_out(self):
        conn, _ = self.lout.accept()
        conn.setblocking(True)
        conn.settimeout(1.0)
        if self.broadcast:
            self.out_all.append(conn)
        else:
            k = min(range(self.S), key=lambda i: (len(self.out_conns[i]), i))
            self.out_conns[k].append(conn)

    def _drop_in(self, sid):
        conn = self.in_conn.pop(sid, None)
        if conn is not None:
            try:
                self.sel.unregister(conn)
            except Exception:
                pass
            conn.close()
        self.rxbuf.pop(sid, None)
        self.asm.fill[sid] = 0

    def _on_data(self, sid):
        conn = self.in_conn.get(sid)
        if conn is None:
            return
        try:
            data = conn.recv(1 << 16)
        except BlockingIOError:
            return
        except OSError:
            data = b""
        if not data:
            self._drop_in(sid)
            return
        self.rxbuf[sid] += data
        self._feed(sid)

    def _feed(self, sid):
        """Move whole 2560-byte packets into the assembler until the frame is full."""
        buf = self.rxbuf[sid]
        P = wire.INPUT_PACKET_BYTES
        room = (self.hop - int(self.asm.fill[sid])) // wire.SAMPLES_PER_PACKET
        n = min(len(buf) // P, room)
        if n > 0:
            self.asm.push(sid, bytes(buf[:n * P]))
            del buf[:n * P]
            if self.asm.fill[sid] == self.hop and self.first_ready_t is None:
                self.first_ready_t = time.time()

    # ---- ticking -------------------------------------------------------------------------------
    def _maybe_tick(self):
        ready = self.asm.ready()
        if len(ready) == 0:
            return
        all_ready = len(ready) == len(self.in_conn)
        waited = self.first_ready_t is not None and time.time() - self.first_ready_t >= self.max_wait_s
        if not (all_ready or waited):
            return
        frames = self.asm.pop(ready)
        echo = self.asm.last_echo
        if self._status_contract:
            res = self.vap.process(frames, ready.astype(np.int32), on_numeric="status")
        else:                                             # a user-supplied model with the two-argument contract
            res = self.vap.process(frames, ready.astype(np.int32))
        t = time.time()
        # a stream whose results are not finite (poisoned state, engine status column) is reset and gets no packet this
        # tick; every other stream of the batch is served as usual
        status = res.get("status")
        bad = set(np.nonzero(status)[0].tolist()) if status is not None else set()
        for k in bad:
            self.vap.reset(int(ready[k]))
            self.numeric_resets += 1
        batch = wire.frame_results_batch(t, echo, res["p_now"], res["p_future"], res["vad"]) if self.mode == "vap" else None
        for k, sid in enumerate(ready):
            sid = int(sid)
            if k in bad:
                continue
            if batch is not None:
                pkt = memoryview(batch[k])
                for conn in list(self.out_all if self.broadcast else self.out_conns[sid]):
                    try:
                        conn.sendall(pkt)
                    except OSError:
                        (self.out_all if self.broadcast else self.out_conns[sid]).remove(conn)
                continue
            r = {"t": t, "x1": echo[k, 0], "x2": echo[k, 1]}
            if self.mode == "bc":
                r.update(p_bc_react=[res["aux"][k, 1]], p_bc_emo=[res["aux"][k, 2]])
            else:
                # p_bc of EVERY window row (vap_nod_main.py:276 indexes the batch dim; :398-406 sends all n values): the
                # engine returns them in the logits columns of the row
                p_bc = res["p_bc"][k] if "p_bc" in res else res["logits"][k, :int(res["n"][k])]
                r.update(p_bc=p_bc, p_nod_short=[res["aux"][k, 1]], p_nod_long=[res["aux"][k, 2]], p_nod_long_p=[res["aux"][k, 3]])
            pkt = wire.frame_result(r, self.mode)
            for conn in list(self.out_all if self.broadcast else self.out_conns[sid]):
                try:
                    conn.sendall(pkt)
                except OSError:
                    (self.out_all if self.broadcast else self.out_conns[sid]).remove(conn)
        self.frames_done += len(ready)
        self.first_ready_t = None
        for sid in list(self.in_conn):        # packets that arrived while the frame was full
            self._feed(sid)

    # ---- main loop -----------------------------------------------------------------------------
    def poll(self, timeout: float = 0.001):
        for key, _ in self.sel.select(timeout):
            kind, sid = key.data
            if kind == "accept_in":
                self._accept_in()
            elif kind == "accept_out":
                self._accept_out()
            else:
                self._on_data(sid)
        self._maybe_tick()

    def serve_forever(self):
        while not self._stop.is_set():
            self.poll()

    def start(self):
        self._thread = threading.Thread(target=self.serve_forever, daemon=True)
        self._thread.start()
        return self

    def stop(self):
        self._stop.set()
        if self._thread:
            self._thread.join(timeout=2)
        for s in [self.lin, self.lout, *self.in_conn.values(), *self.out_all, *[c for l in self.out_conns for c in l]]:
            try:
                s.close()
            except Exception:
                pass
        self.sel.close()
