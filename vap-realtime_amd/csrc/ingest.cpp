// Native many-stream TCP front-end of libvapx (include/vapx.h, "vapx_ingest_*"), host code only.
//
// Reference being replaced: proc_serv_in / proc_serv_out / proc_serv_out_dist, rvap/vap_main/vap_main.py:338-457, and the
// per-sample Python codec rvap/common/util.py:52-237 — one dialogue per process there, thousands per process here.
//
//   rx threads (epoll)      recv -> decode f64 pairs -> per-stream frame buffers (f32 staging in page-locked memory + the
//                           f64 echo the result packet carries); a complete frame is queued for the tick thread
//   tick thread             batches the ready frames (ragged), vapx_step (host in / host out), hands rows to the senders
//   tx threads              encode header / tail, sendmsg() with the echo arrays as iovecs (no copy), free the frame buffer
//
// Every stream owns NBUF frame buffers (filling / waiting / in flight / being sent + slack), so reception never waits for the GPU
// unless a sender outruns the engine by four whole frames; then the connection is paused (TCP back-pressure), never dropped.
#include <arpa/inet.h>
#include <errno.h>
#include <fcntl.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <pthread.h>
#include <sched.h>
#include <sys/epoll.h>
#include <sys/eventfd.h>
#include <sys/resource.h>
#include <sys/socket.h>
#include <sys/uio.h>
#include <sys/un.h>
#include <poll.h>
#include <time.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstddef>
#include <cstdio>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/vapx.h"

namespace {

constexpr int NBUF = 5;   // frame buffers per stream: filling + waiting + in flight + two of slack for a client that bursts after a stall
constexpr int PAIR_BYTES = 16;                 // one sample of both channels: f64 ch1, f64 ch2 (util.py:52-62)
enum BufState : int { B_FREE = 0, B_FILLING = 1, B_READY = 2, B_INFLIGHT = 3 };

// Timed condition wait.  Under ThreadSanitizer the wait goes through system_clock (pthread_cond_timedwait): gcc-11's libtsan does
// not intercept pthread_cond_clockwait, loses the unlock inside it and then reports every access under that mutex as a race.
template <class Rep, class Period>
inline void cv_wait_for(std::condition_variable& cv, std::unique_lock<std::mutex>& lk, std::chrono::duration<Rep, Period> d) {
#if defined(__SANITIZE_THREAD__)
  cv.wait_until(lk, std::chrono::system_clock::now() + d);
#else
  cv.wait_for(lk, d);
#endif
}

double mono_now() {
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
double unix_now() {
  timespec ts;
  clock_gettime(CLOCK_REALTIME, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

inline void atomic_max(std::atomic<int64_t>& a, int64_t v) {
  int64_t m = a.load(std::memory_order_relaxed);
  while (v > m && !a.compare_exchange_weak(m, v, std::memory_order_relaxed)) {}
}

// ---- codec ---------------------------------------------------------------------------------------
inline void put_u32(uint8_t*& p, uint32_t v) { memcpy(p, &v, 4); p += 4; }   // little-endian host (x86-64 / the GPU box)
inline void put_f64(uint8_t*& p, double v) { memcpy(p, &v, 8); p += 8; }

// number of head values after the echo blocks, by mode
int tail_bytes(int mode, int n_rows_pbc) {
  if (mode == VAPX_MODE_VAP) return 3 * (4 + 16);
  if (mode == VAPX_MODE_BC) return 2 * (4 + 8);
  return (4 + 8 * n_rows_pbc) + 3 * (4 + 8);
}

// heads of one out row -> tail of the packet (after "u32 n | x2"); returns bytes written
int encode_tail(int mode, const float* row, uint8_t* dst) {
  uint8_t* p = dst;
  if (mode == VAPX_MODE_VAP) {            // u32 2 | p_now | u32 2 | p_future | u32 2 | vad   (util.py:134-141)
    for (int blk = 0; blk < 3; ++blk) {
      put_u32(p, 2);
      put_f64(p, (double)row[blk * 2]);
      put_f64(p, (double)row[blk * 2 + 1]);
    }
  } else if (mode == VAPX_MODE_BC) {      // u32 1 | p_bc_react | u32 1 | p_bc_emo            (util.py:193-211)
    put_u32(p, 1); put_f64(p, (double)row[VAPX_OUT_AUX + 1]);
    put_u32(p, 1); put_f64(p, (double)row[VAPX_OUT_AUX + 2]);
  } else {                                // u32 n | p_bc rows | 3 x (u32 1 | p)              (util.py:213-237, vap_nod_main.py:276)
    // clamped to the tail buffer's 256 rows: a step function of vapx_ingest_open_fn that leaves the column unset must not
    // turn into an out-of-bounds write here (vapx_wire_encode_result validates the same range)
    int n = (int)row[VAPX_OUT_NVALID];
    n = n < 0 ? 0 : (n > 256 ? 256 : n);
    put_u32(p, (uint32_t)n);
    for (int i = 0; i < n; ++i) put_f64(p, (double)row[VAPX_OUT_LOGITS + i]);
    for (int k = 1; k <= 3; ++k) { put_u32(p, 1); put_f64(p, (double)row[VAPX_OUT_AUX + k]); }
  }
  return (int)(p - dst);
}

struct Hist {   // log2-spaced latency histogram, 8 sub-buckets per octave from 1 us
  static constexpr int N = 8 * 27;
  std::atomic<int64_t> b[N];
  std::atomic<int64_t> cnt{0};
  std::atomic<int64_t> sum_us{0};
  std::atomic<int64_t> max_us{0};
  Hist() { for (auto& x : b) x = 0; }
  void add(double sec) {
    double us = sec * 1e6;
    if (us < 1.0) us = 1.0;
    int k = (int)(log2(us) * 8.0);
    if (k >= N) k = N - 1;
    b[k].fetch_add(1, std::memory_order_relaxed);
    cnt.fetch_add(1, std::memory_order_relaxed);
    sum_us.fetch_add((int64_t)us, std::memory_order_relaxed);
    int64_t m = max_us.load(std::memory_order_relaxed), v = (int64_t)us;
    while (v > m && !max_us.compare_exchange_weak(m, v, std::memory_order_relaxed)) {}
  }
  double pct(double q) const {
    int64_t total = cnt.load(), acc = 0;
    if (total == 0) return 0.0;
    int64_t want = (int64_t)ceil(q * (double)total);
    for (int k = 0; k < N; ++k) {
      acc += b[k].load();
      if (acc >= want) return exp2((k + 1) / 8.0) * 1e-3;   // upper edge of the bucket, ms
    }
    return (double)max_us.load() * 1e-3;
  }
  void clear() { for (auto& x : b) x = 0; cnt = 0; sum_us = 0; max_us = 0; }
};

struct Slot {
  int fd_in = -1;
  std::atomic<uint32_t> gen{0};           // bumped per connection: queued frames of a dead connection are dropped (read by the tick thread)
  std::atomic<int> state[NBUF];
  double t_ready[NBUF] = {};
  int wbuf = -1, fill = 0;                // rx thread only
  uint8_t partial[PAIR_BYTES]; int npartial = 0;
  std::vector<uint8_t> backlog;           // bytes received while no buffer was free
  int lowat = 0;                          // rx thread only: SO_RCVLOWAT currently set on fd_in (0 = the kernel default)
  std::atomic<bool> paused{false};        // written by the slot's rx / accept thread, read by whoever frees a buffer
  std::mutex lmu;                         // guards listeners
  std::vector<int> listeners;
  // VAPX_INGEST_DEBUG bookkeeping (relaxed counters, printed at close): where do a stream's frames go?
  std::atomic<int64_t> dbg_ready{0}, dbg_sent{0}, dbg_nolistener{0};
  double dbg_t_first_ready = 0, dbg_t_attach = 0, dbg_t_first_sent = 0;
  Slot() { for (auto& s : state) s = B_FREE; }
};

struct Ready { int slot; int buf; uint32_t gen; double t; };

struct Job {                               // one tick's rows on their way out
  int n = 0;
  std::vector<Ready> rows;
  float* out = nullptr;                    // [max_batch][VAPX_OUT_STRIDE] pinned
  double t_unix = 0;
  std::vector<std::vector<int>> part;      // row indices per sender thread: slot % X — a stream is always sent by the same thread
  int senders_done = 0;                    // sender threads that are through with this job (under job_mu)
  bool busy = false;
};

}  // namespace

struct vapx_ingest {
  vapx_ingest_config cfg;
  vapx_ingest_step_fn step = nullptr;
  vapx_ingest_reset_fn reset = nullptr;
  void* user = nullptr;
  vapx_handle engine = nullptr;
  int S = 0, max_batch = 0, hop = 0, mode = 0, hz = 0;
  bool broadcast = false;
  int R = 2, X = 2;

  std::unique_ptr<Slot[]> slots;
  float* stage = nullptr;                  // pinned [S][NBUF][2][hop] f32
  std::vector<double> echo;                // [S][NBUF][2][hop] f64
  float* batch_audio = nullptr;            // pinned [max_batch][2][hop]
  std::vector<int32_t> batch_ids;
  Job jobs[2];

  int lin = -1, lout = -1;
  int port_in = 0, port_out = 0;
  std::vector<int> ep;                     // epoll fd per rx thread
  std::vector<int> wake;                   // eventfd per rx thread (resume / stop)
  std::vector<std::thread> rx_threads, tx_threads;
  std::thread tick_thread;
  std::atomic<bool> stop{false};

  std::mutex slots_mu;                     // slot allocation, out_all, listener bookkeeping
  std::vector<int> out_all;                // broadcast listeners
  std::vector<int> lcount;                 // listeners per slot (under slots_mu)
  int lmin = 0, lcursor = 0;               // fewest listeners on any slot; lowest slot that may still have that few
  int free_hint = 0;                       // no input slot below this one is free (under slots_mu): adoption is O(1) amortised, not a scan of S slots
  int ep_accept = -1;                      // the accept thread's epoll (listen sockets only)
  std::thread accept_thread;
  std::mutex ready_mu;
  std::condition_variable ready_cv;
  std::vector<Ready> ready;                // rx -> tick
  std::vector<std::pair<int, int>> resets; // (slot, carry_only) rx -> tick, under ready_mu
  std::vector<std::mutex> resume_mu;
  std::vector<std::vector<int>> resume;    // tick/tx -> rx: slots with a free buffer again

  std::mutex job_mu;
  std::condition_variable job_cv, job_done_cv;
  int64_t jobs_published = 0;              // job k of the run lives in jobs[k & 1] (under job_mu)

  // stats
  std::atomic<int64_t> frames_done{0}, ticks{0}, rx_bytes{0}, tx_bytes{0}, in_conns{0}, out_conns{0}, dropped{0}, numeric_resets{0},
      overruns{0}, batch_sum{0};
  std::atomic<int64_t> step_us{0};
  // stall diagnostics (printed at close when VAPX_INGEST_DEBUG is set): longest single pass of an rx thread over its ready sockets, longest gap
  // between two passes that both had work, longest step
  std::atomic<int64_t> dbg_rx_pass_us{0}, dbg_rx_gap_us{0}, dbg_step_us{0}, dbg_send_us{0}, dbg_job_tx_us{0}, dbg_recv_us{0};
  double dbg_t_in_first = 0, dbg_t_in_last = 0, dbg_t_out_first = 0, dbg_t_out_last = 0;   // accept thread only: CLOCK_MONOTONIC of the first / last adopted connection
  std::atomic<int64_t> accept_fd_errors{0};   // accept4() failed for lack of descriptors (EMFILE / ENFILE): the process limit is below 2 x streams
  Hist lat;
  std::atomic<int64_t> late10{0};          // packets of the current latency window handed over > 10 ms after their frame was complete
  bool no_lowat = false;                   // VAPX_INGEST_NO_LOWAT set at open: wake per packet as rounds 1-4 did (A/B measurements only)
  bool debug = false;                      // VAPX_INGEST_DEBUG set at open: per-call stall diagnostics (two clock reads per recv / sendmsg)
  std::string err;
  bool pinned_blocks = true;               // staging came from vapx_host_alloc (false: plain calloc, no HIP device)
  // link to a front door in ANOTHER process (vapx_ingest_attach_link): accepted connections arrive as descriptors over a unix socket,
  // slot releases / listener drops are reported back
  int link_fd = -1;
  std::mutex link_mu;                      // serialises the sends (rx / tx / link threads)
  std::thread link_thread;

  float* f32buf(int slot, int buf) { return stage + ((size_t)slot * NBUF + buf) * 2 * hop; }
  double* f64buf(int slot, int buf) { return echo.data() + ((size_t)slot * NBUF + buf) * 2 * hop; }
};

namespace {

int listen_on(int port, bool any, int* bound) {
  int s = socket(AF_INET, SOCK_STREAM | SOCK_NONBLOCK, 0);
  if (s < 0) return -1;
  int one = 1;
  setsockopt(s, SOL_SOCKET, SO_REUSEADDR, &one, sizeof one);
  sockaddr_in a;
  memset(&a, 0, sizeof a);
  a.sin_family = AF_INET;
  a.sin_addr.s_addr = htonl(any ? INADDR_ANY : INADDR_LOOPBACK);
  a.sin_port = htons((uint16_t)port);
  if (bind(s, (sockaddr*)&a, sizeof a) != 0 || listen(s, 8192) != 0) { close(s); return -1; }
  socklen_t len = sizeof a;
  getsockname(s, (sockaddr*)&a, &len);
  *bound = ntohs(a.sin_port);
  return s;
}

// tags in epoll_event.data.u64: low 32 bits slot (or a listen-socket / wake id), high bits kind
constexpr uint64_t K_DATA = 0, K_LISTEN_IN = 1ull << 32, K_LISTEN_OUT = 2ull << 32, K_WAKE = 3ull << 32;

void ep_add(int ep, int fd, uint64_t tag) {
  epoll_event ev;
  memset(&ev, 0, sizeof ev);
  ev.events = EPOLLIN;
  ev.data.u64 = tag;
  epoll_ctl(ep, EPOLL_CTL_ADD, fd, &ev);
}
void ep_mod(int ep, int fd, uint64_t tag, bool want_in) {
  epoll_event ev;
  memset(&ev, 0, sizeof ev);
  ev.events = want_in ? EPOLLIN : 0;
  ev.data.u64 = tag;
  epoll_ctl(ep, EPOLL_CTL_MOD, fd, &ev);
}

void kick(int efd) {
  uint64_t one = 1;
  ssize_t r = write(efd, &one, 8);
  (void)r;
}

int pick_free(Slot& s) {
  for (int b = 0; b < NBUF; ++b) {
    int expect = B_FREE;
    if (s.state[b].compare_exchange_strong(expect, B_FILLING)) return b;
  }
  return -1;
}

// free a frame buffer and, if its stream was paused for lack of buffers, ask its rx thread to resume it
void release_buf(vapx_ingest* g, int slot, int buf) {
  g->slots[slot].state[buf].store(B_FREE, std::memory_order_release);
  const int r = slot % g->R;
  bool need = false;
  {
    std::lock_guard<std::mutex> lk(g->resume_mu[r]);
    // a stale read only costs a spurious wake-up
    if (g->slots[slot].paused.load(std::memory_order_acquire)) { g->resume[r].push_back(slot); need = true; }
  }
  if (need) kick(g->wake[r]);
}

// ---- link protocol (front-door process <-> worker process), one SOCK_SEQPACKET unix socket per worker ----
struct LinkMsg { int32_t kind, a, b, c; };
enum { LK_HELLO = 1,      // worker -> door: a = dialogue slots, b = frame_hz, c = mode | broadcast << 8
       LK_ADOPT_IN = 2,   // door -> worker: a = local slot, + the accepted input socket (SCM_RIGHTS)
       LK_ADOPT_OUT = 3,  // door -> worker: a = local slot the listener attaches to (ignored by a broadcast shard), + the socket
       LK_IN_CLOSED = 4,  // worker -> door: the input connection of slot a is gone (the slot is free again)
       LK_OUT_CLOSED = 5  // worker -> door: a listener of slot a was dropped (broadcast shard: a = -1)
};

bool link_send(int link, std::mutex* mu, const LinkMsg& m, int pass_fd = -1) {
  msghdr mh;
  memset(&mh, 0, sizeof mh);
  iovec iov{(void*)&m, sizeof m};
  mh.msg_iov = &iov; mh.msg_iovlen = 1;
  alignas(cmsghdr) char ctl[CMSG_SPACE(sizeof(int))];
  if (pass_fd >= 0) {
    memset(ctl, 0, sizeof ctl);
    mh.msg_control = ctl; mh.msg_controllen = sizeof ctl;
    cmsghdr* c = CMSG_FIRSTHDR(&mh);
    c->cmsg_level = SOL_SOCKET; c->cmsg_type = SCM_RIGHTS; c->cmsg_len = CMSG_LEN(sizeof(int));
    memcpy(CMSG_DATA(c), &pass_fd, sizeof(int));
  }
  std::unique_lock<std::mutex> lk;
  if (mu) lk = std::unique_lock<std::mutex>(*mu);
  for (;;) {
    ssize_t w = sendmsg(link, &mh, MSG_NOSIGNAL);
    if (w == (ssize_t)sizeof m) return true;
    if (w < 0 && (errno == EINTR)) continue;
    if (w < 0 && (errno == EAGAIN || errno == EWOULDBLOCK)) { pollfd p{link, POLLOUT, 0}; poll(&p, 1, 50); continue; }
    return false;
  }
}

// one message (and at most one descriptor); returns 1, 0 = nothing within `timeout_ms`, -1 = the peer is gone
int link_recv(int link, LinkMsg* m, int* got_fd, int timeout_ms) {
  pollfd p{link, POLLIN, 0};
  const int pr = poll(&p, 1, timeout_ms);
  if (pr == 0) return 0;
  if (pr < 0) return errno == EINTR ? 0 : -1;
  msghdr mh;
  memset(&mh, 0, sizeof mh);
  iovec iov{m, sizeof *m};
  mh.msg_iov = &iov; mh.msg_iovlen = 1;
  alignas(cmsghdr) char ctl[CMSG_SPACE(sizeof(int))];
  mh.msg_control = ctl; mh.msg_controllen = sizeof ctl;
  ssize_t r = recvmsg(link, &mh, MSG_CMSG_CLOEXEC | MSG_DONTWAIT);
  if (r < 0) return (errno == EAGAIN || errno == EWOULDBLOCK || errno == EINTR) ? 0 : -1;
  if (r == 0) return -1;
  if (got_fd) {
    *got_fd = -1;
    for (cmsghdr* c = CMSG_FIRSTHDR(&mh); c; c = CMSG_NXTHDR(&mh, c))
      if (c->cmsg_level == SOL_SOCKET && c->cmsg_type == SCM_RIGHTS) memcpy(got_fd, CMSG_DATA(c), sizeof(int));
  }
  return r == (ssize_t)sizeof *m ? 1 : -1;
}

void link_event(vapx_ingest* g, int kind, int slot) {
  if (g->link_fd >= 0) link_send(g->link_fd, &g->link_mu, LinkMsg{kind, slot, 0, 0});
}

void drop_input(vapx_ingest* g, int r, int slot) {
  Slot& s = g->slots[slot];
  if (s.fd_in < 0) return;
  epoll_ctl(g->ep[r], EPOLL_CTL_DEL, s.fd_in, nullptr);
  close(s.fd_in);
  if (s.wbuf >= 0) { s.state[s.wbuf].store(B_FREE); s.wbuf = -1; }
  s.fill = 0; s.npartial = 0; s.backlog.clear(); s.paused = false;
  {
    std::lock_guard<std::mutex> lk(g->slots_mu);
    s.fd_in = -1;
    if (slot < g->free_hint) g->free_hint = slot;
    s.gen.fetch_add(1);                     // frames of this connection still queued are dropped by the tick thread
  }
  link_event(g, LK_IN_CLOSED, slot);       // (before the counter: who sees the count drop finds the door's mirror told already)
  g->in_conns.fetch_sub(1);
}

// decode `n` bytes of the stream into the slot's frame buffers; returns bytes consumed (< n: no free buffer)
size_t feed(vapx_ingest* g, int slot, const uint8_t* p, size_t n) {
  Slot& s = g->slots[slot];
  const int hop = g->hop;
  const double gain = g->cfg.gain;
  size_t used = 0;
  while (used < n) {
    if (s.wbuf < 0) {
      s.wbuf = pick_free(s);
      if (s.wbuf < 0) return used;          // engine is two frames behind this sender
      s.fill = 0;
    }
    if (s.npartial) {                       // finish a sample pair split across two reads
      size_t take = std::min<size_t>(PAIR_BYTES - s.npartial, n - used);
      memcpy(s.partial + s.npartial, p + used, take);
      s.npartial += (int)take; used += take;
      if (s.npartial < PAIR_BYTES) break;
      double v[2];
      memcpy(v, s.partial, PAIR_BYTES);
      s.npartial = 0;
      float* f = g->f32buf(slot, s.wbuf);
      double* d = g->f64buf(slot, s.wbuf);
      const double a = gain != 1.0 ? v[0] * gain : v[0], b = gain != 1.0 ? v[1] * gain : v[1];
      d[s.fill] = a; d[hop + s.fill] = b;
      f[s.fill] = (float)a; f[hop + s.fill] = (float)b;
      ++s.fill;
    } else {
      size_t pairs = std::min<size_t>((n - used) / PAIR_BYTES, (size_t)(hop - s.fill));
      if (pairs == 0) {                     // fewer than 16 bytes left: keep them for the next read
        size_t rest = n - used;
        if (rest >= (size_t)PAIR_BYTES) { /* frame full, handled below */ }
        else { memcpy(s.partial, p + used, rest); s.npartial = (int)rest; used = n; break; }
      }
      float* f = g->f32buf(slot, s.wbuf) + s.fill;
      double* d = g->f64buf(slot, s.wbuf) + s.fill;
      const uint8_t* q = p + used;
      if (gain != 1.0) {
        for (size_t i = 0; i < pairs; ++i) {
          double v[2];
          memcpy(v, q + i * PAIR_BYTES, PAIR_BYTES);
          const double a = v[0] * gain, b = v[1] * gain;     // float64 multiply BEFORE the f32 cast (vap_main.py:393-395)
          d[i] = a; d[hop + i] = b;
          f[i] = (float)a; f[hop + i] = (float)b;
        }
      } else {
        for (size_t i = 0; i < pairs; ++i) {
          double v[2];
          memcpy(v, q + i * PAIR_BYTES, PAIR_BYTES);
          d[i] = v[0]; d[hop + i] = v[1];
          f[i] = (float)v[0]; f[hop + i] = (float)v[1];       // == torch.tensor(float64).float() (vap_main.py:266-270)
        }
      }
      s.fill += (int)pairs;
      used += pairs * PAIR_BYTES;
    }
    if (s.fill == hop) {                    // frame complete -> tick thread
      const int b = s.wbuf;
      const double t = mono_now();
      s.t_ready[b] = t;
      if (s.dbg_ready.fetch_add(1, std::memory_order_relaxed) == 0) s.dbg_t_first_ready = t;
      s.state[b].store(B_READY, std::memory_order_release);
      s.wbuf = -1; s.fill = 0;
      {
        std::lock_guard<std::mutex> lk(g->ready_mu);
        g->ready.push_back({slot, b, s.gen.load(std::memory_order_relaxed), t});
      }
      g->ready_cv.notify_one();
    }
  }
  return used;
}

// Wake this socket's receive thread when the REST OF THE CURRENT FRAME is there, not for every 10 ms packet of it (SO_RCVLOWAT; epoll
// honours it): a client sends hop / 160 packets per frame (vap_main.py:373-391) and nothing can be done with a part of a frame, so 4096
// dialogues cost 82 k wake-ups + recv() calls per second instead of 410 k.  Less time on the host's cores is less exposure to whoever else
// runs there (round 5: on a box with load average 60 from other tenants every front-end thread was losing ~10 ms slices).  Re-armed only
// when the value changes: a sender that delivers whole frames never causes a setsockopt.
void arm_lowat(vapx_ingest* g, Slot& s) {
  long need = (long)(g->hop - (s.wbuf >= 0 ? s.fill : 0)) * (long)PAIR_BYTES - s.npartial;
  if (need < 1) need = 1;
  if (need > 65536) need = 65536;           // (5 Hz frames are 51 KB; stay well inside the receive buffer)
  if ((int)need == s.lowat || s.fd_in < 0 || g->no_lowat) return;
  int v = (int)need;
  if (setsockopt(s.fd_in, SOL_SOCKET, SO_RCVLOWAT, &v, sizeof v) == 0) s.lowat = v;
}

void on_data(vapx_ingest* g, int r, int slot, uint8_t* scratch, size_t cap, uint32_t events) {
  Slot& s = g->slots[slot];
  if (s.fd_in < 0) return;
  if (s.paused) {
    // a paused slot is armed with events = 0, but EPOLLHUP / EPOLLERR are reported regardless (level-triggered): never read
    // new bytes past the parked backlog (they would overtake it), and a peer that is gone is dropped instead of spinning here
    if (events & (EPOLLHUP | EPOLLERR)) drop_input(g, r, slot);
    return;
  }
  for (int round = 0; round < 4; ++round) {   // bounded work per wake-up: fairness between streams
    const double tr0 = g->debug ? mono_now() : 0.0;
    ssize_t n = recv(s.fd_in, scratch, cap, 0);
    if (g->debug) atomic_max(g->dbg_recv_us, (int64_t)((mono_now() - tr0) * 1e6));   // longest recv() call
    if (n < 0) {
      if (errno == EAGAIN || errno == EWOULDBLOCK || errno == EINTR) { arm_lowat(g, s); return; }
      drop_input(g, r, slot);
      return;
    }
    if (n == 0) { drop_input(g, r, slot); return; }
    g->rx_bytes.fetch_add(n, std::memory_order_relaxed);
    size_t used = feed(g, slot, scratch, (size_t)n);
    if (used < (size_t)n) {                 // no free frame buffer: park the rest and stop reading this socket
      s.backlog.insert(s.backlog.end(), scratch + used, scratch + n);   // (append: never overwrite bytes parked earlier)
      {
        std::lock_guard<std::mutex> lk(g->resume_mu[r]);
        s.paused = true;
      }
      g->overruns.fetch_add(1);
      ep_mod(g->ep[r], s.fd_in, K_DATA | (uint32_t)slot, false);   // (SO_RCVLOWAT is re-armed by do_resume, from the state the backlog leaves)
      // a buffer may have been released between pick_free and `paused = true`
      for (int b = 0; b < NBUF; ++b)
        if (s.state[b].load() == B_FREE) {
          std::lock_guard<std::mutex> lk(g->resume_mu[r]);
          g->resume[r].push_back(slot);
          kick(g->wake[r]);
          break;
        }
      return;
    }
    if ((size_t)n < cap) { arm_lowat(g, s); return; }
  }
  arm_lowat(g, s);
}

void do_resume(vapx_ingest* g, int r, int slot) {
  Slot& s = g->slots[slot];
  if (s.fd_in < 0 || !s.paused) return;
  if (!s.backlog.empty()) {
    std::vector<uint8_t> b;
    b.swap(s.backlog);
    size_t used = feed(g, slot, b.data(), b.size());
    if (used < b.size()) { s.backlog.assign(b.begin() + used, b.end()); return; }   // still no buffer: stay paused
  }
  {
    std::lock_guard<std::mutex> lk(g->resume_mu[r]);
    s.paused = false;
  }
  // The backlog moved `fill` / `npartial`: the socket's SO_RCVLOWAT still holds what the frame open BEFORE the pause lacked, and the rest
  // of the frame open NOW may be fewer bytes than that - they would sit in the socket unread (a finite stream lost its last frame, a
  // live one got it a frame period late).  Set the threshold first, then re-enable EPOLLIN: EPOLL_CTL_MOD polls the socket against it.
  arm_lowat(g, s);
  ep_mod(g->ep[r], s.fd_in, K_DATA | (uint32_t)slot, true);
}

// accept4 failed: out of descriptors -> count it and back off (the listen socket stays readable, the level-triggered epoll would spin)
bool accept_failed_for_fds(vapx_ingest* g) {
  if (errno != EMFILE && errno != ENFILE && errno != ENOBUFS && errno != ENOMEM) return false;
  if (g->accept_fd_errors.fetch_add(1) == 0)
    fprintf(stderr, "[vapx ingest] accept: out of file descriptors (%s) - raise `ulimit -n` above 2 x streams + 64; connections wait in the listen queue\n", strerror(errno));
  std::this_thread::sleep_for(std::chrono::milliseconds(10));
  return true;
}

// lowest free input slot, or -1 (the front door compares this across shards)
int lowest_free_slot(vapx_ingest* g) {
  std::lock_guard<std::mutex> lk(g->slots_mu);
  for (int i = g->free_hint; i < g->S; ++i)
    if (g->slots[i].fd_in < 0) { g->free_hint = i; return i; }
  g->free_hint = g->S;
  return -1;
}

// take over an accepted input connection: the lowest free stream slot gets it; false = every slot is taken (fd untouched)
bool adopt_in(vapx_ingest* g, int fd) {
  int slot = -1;
  {
    // (round 4: this was a scan from slot 0 — 4096 dialogues connecting at once cost the accept thread ~S^2 / 2 slot visits, the output
    // connections queued behind them were attached after the first frames had been answered, and those answers went to nobody)
    std::lock_guard<std::mutex> lk(g->slots_mu);
    for (int i = g->free_hint; i < g->S; ++i)
      if (g->slots[i].fd_in < 0) { slot = i; break; }
    g->free_hint = slot >= 0 ? slot + 1 : g->S;
    if (slot >= 0) g->slots[slot].fd_in = fd;
  }
  if (slot < 0) return false;
  int one = 1;
  setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof one);
  Slot& s = g->slots[slot];
  s.wbuf = -1; s.fill = 0; s.npartial = 0; s.backlog.clear(); s.paused = false; s.lowat = 0;
  {
    // the carry restarts from zeros for every connection (vap_main.py:368-369); reset_on_connect also clears the
    // model state, which the reference keeps.  Applied by the tick thread before its next step.
    std::lock_guard<std::mutex> lk(g->ready_mu);
    g->resets.push_back({slot, g->cfg.reset_on_connect ? 0 : 1});
  }
  g->in_conns.fetch_add(1);
  { const double t = mono_now(); if (g->dbg_t_in_first == 0) g->dbg_t_in_first = t; g->dbg_t_in_last = t; }
  ep_add(g->ep[slot % g->R], fd, K_DATA | (uint32_t)slot);
  return true;
}

void accept_in(vapx_ingest* g) {
  for (;;) {
    int fd = accept4(g->lin, nullptr, nullptr, SOCK_NONBLOCK);
    if (fd < 0) { accept_failed_for_fds(g); return; }
    if (!adopt_in(g, fd)) close(fd);        // every stream slot is taken
  }
}

// the (listener count, slot) an output connection attached now would get: fewest listeners, lowest index first
// (amortised O(1): a cursor walks the slots that still have `lmin` listeners and wraps with lmin + 1)
std::pair<int, int> next_listener_slot(vapx_ingest* g) {
  std::lock_guard<std::mutex> lk(g->slots_mu);
  for (int pass = 0; pass < 2; ++pass) {
    for (int i = g->lcursor; i < g->S; ++i)
      if (g->lcount[i] == g->lmin) { g->lcursor = i; return {g->lmin, i}; }
    ++g->lmin; g->lcursor = 0;
  }
  return {g->lmin, 0};
}

// take over an accepted output connection (non-blocking like vap_main.py:346-347): the k-th output connection hears the k-th stream
void adopt_out(vapx_ingest* g, int fd) {
  int one = 1;
  setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof one);
  g->out_conns.fetch_add(1);
  { const double t = mono_now(); if (g->dbg_t_out_first == 0) g->dbg_t_out_first = t; g->dbg_t_out_last = t; }
  if (g->broadcast) { std::lock_guard<std::mutex> lk(g->slots_mu); g->out_all.push_back(fd); return; }
  const int best = next_listener_slot(g).second;
  std::lock_guard<std::mutex> lk(g->slots_mu);
  g->lcursor = best + 1;
  ++g->lcount[best];
  std::lock_guard<std::mutex> l2(g->slots[best].lmu);
  g->slots[best].listeners.push_back(fd);
  if (g->slots[best].dbg_t_attach == 0) g->slots[best].dbg_t_attach = mono_now();
}

void accept_out(vapx_ingest* g) {
  for (;;) {
    int fd = accept4(g->lout, nullptr, nullptr, SOCK_NONBLOCK);
    if (fd < 0) { accept_failed_for_fds(g); return; }
    adopt_out(g, fd);
  }
}

// a dropped listener lowers its slot's count: the next output connection goes there first
void listener_dropped(vapx_ingest* g, int slot) {
  {
    std::lock_guard<std::mutex> lk(g->slots_mu);
    if (--g->lcount[slot] < g->lmin) { g->lmin = g->lcount[slot]; g->lcursor = slot; }
    else if (g->lcount[slot] == g->lmin && slot < g->lcursor) g->lcursor = slot;
  }
  link_event(g, LK_OUT_CLOSED, slot);
}

// the same adoptions with the slot named by a front door in another process (its mirror of this shard is the allocator)
bool adopt_in_at(vapx_ingest* g, int fd, int slot) {
  if (slot < 0 || slot >= g->S) return false;
  {
    std::lock_guard<std::mutex> lk(g->slots_mu);
    if (g->slots[slot].fd_in >= 0) return false;
    g->slots[slot].fd_in = fd;
  }
  int one = 1;
  setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof one);
  const int fl = fcntl(fd, F_GETFL, 0);
  if (fl >= 0) fcntl(fd, F_SETFL, fl | O_NONBLOCK);
  Slot& s = g->slots[slot];
  s.wbuf = -1; s.fill = 0; s.npartial = 0; s.backlog.clear(); s.paused = false; s.lowat = 0;
  {
    std::lock_guard<std::mutex> lk(g->ready_mu);
    g->resets.push_back({slot, g->cfg.reset_on_connect ? 0 : 1});
  }
  g->in_conns.fetch_add(1);
  ep_add(g->ep[slot % g->R], fd, K_DATA | (uint32_t)slot);
  return true;
}

void adopt_out_at(vapx_ingest* g, int fd, int slot) {
  int one = 1;
  setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof one);
  const int fl = fcntl(fd, F_GETFL, 0);
  if (fl >= 0) fcntl(fd, F_SETFL, fl | O_NONBLOCK);
  g->out_conns.fetch_add(1);
  if (g->broadcast) { std::lock_guard<std::mutex> lk(g->slots_mu); g->out_all.push_back(fd); return; }
  if (slot < 0 || slot >= g->S) slot = 0;
  std::lock_guard<std::mutex> lk(g->slots_mu);
  ++g->lcount[slot];
  std::lock_guard<std::mutex> l2(g->slots[slot].lmu);
  g->slots[slot].listeners.push_back(fd);
}

void link_main(vapx_ingest* g) {
  while (!g->stop.load()) {
    LinkMsg m;
    int fd = -1;
    const int rc = link_recv(g->link_fd, &m, &fd, 100);
    if (rc == 0) continue;
    if (rc < 0) break;                       // the front door is gone: keep serving the dialogues we have, take no new ones
    if (m.kind == LK_ADOPT_IN) {
      if (fd >= 0 && !adopt_in_at(g, fd, m.a)) { close(fd); link_event(g, LK_IN_CLOSED, m.a); }   // (cannot happen while the door's mirror is right)
    } else if (m.kind == LK_ADOPT_OUT) {
      if (fd >= 0) adopt_out_at(g, fd, m.a);
    } else if (fd >= 0) close(fd);
  }
}

void accept_main(vapx_ingest* g) {
  epoll_event evs[8];
  while (!g->stop.load()) {
    int n = epoll_wait(g->ep_accept, evs, 8, 100);
    for (int i = 0; i < n; ++i) {
      const uint64_t kind = evs[i].data.u64 & ~0xffffffffull;
      if (kind == K_LISTEN_IN) accept_in(g);
      else if (kind == K_LISTEN_OUT) accept_out(g);
    }
  }
}


void rx_main(vapx_ingest* g, int r) {
  std::vector<uint8_t> scratch(256 * 1024);
  epoll_event evs[256];
  double t_last_busy = 0.0;
  while (!g->stop.load()) {
    int n = epoll_wait(g->ep[r], evs, 256, 100);
    const double t_in = g->debug ? mono_now() : 0.0;
    if (g->debug && n > 0 && t_last_busy > 0.0) atomic_max(g->dbg_rx_gap_us, (int64_t)((t_in - t_last_busy) * 1e6));
    for (int i = 0; i < n; ++i) {
      const uint64_t tag = evs[i].data.u64, kind = tag & ~0xffffffffull;
      if (kind == K_WAKE) {
        uint64_t v;
        ssize_t rr = read(g->wake[r], &v, 8);
        (void)rr;
        std::vector<int> todo;
        {
          std::lock_guard<std::mutex> lk(g->resume_mu[r]);
          todo.swap(g->resume[r]);
        }
        for (int slot : todo) do_resume(g, r, slot);
      } else on_data(g, r, (int)(tag & 0xffffffffu), scratch.data(), scratch.size(), evs[i].events);
    }
    if (g->debug) {
      if (n > 0) {
        t_last_busy = mono_now();
        atomic_max(g->dbg_rx_pass_us, (int64_t)((t_last_busy - t_in) * 1e6));
      } else t_last_busy = 0.0;
    }
  }
}

// send one result packet to one listener; false = the listener could not take it (caller drops it)
bool send_packet(int fd, iovec* iov, int niov, size_t total) {
  msghdr mh;
  memset(&mh, 0, sizeof mh);
  mh.msg_iov = iov;
  mh.msg_iovlen = niov;
  ssize_t n = sendmsg(fd, &mh, MSG_NOSIGNAL | MSG_DONTWAIT);
  return n == (ssize_t)total;
}

void tx_main(vapx_ingest* g, int x) {
  std::vector<uint8_t> tail(64 + 8 * 256);
  const int hop = g->hop;
  int64_t next = 0;                          // the job this thread sends next: every sender walks over EVERY job, in publication order
  while (true) {
    // Per-stream order without serialising the ticks: the rows of a job are partitioned by slot % X, so a stream's packets always leave
    // through the same thread, and that thread works through the jobs in the order they were published — a listener can never see frame
    // k + 1 before frame k (round 4 kept the order by sending one job at a time: every sender idled while the last chunk of a tick was
    // still going out; advisor r04).  A job is handed back to the tick thread when all X senders are through with it.
    int j = -1;
    {
      std::unique_lock<std::mutex> lk(g->job_mu);
      g->job_cv.wait(lk, [&] { return g->jobs_published > next || g->stop.load(); });
      if (g->jobs_published <= next) return;   // stopping, nothing left to send
      j = (int)(next & 1);
    }
    Job& job = g->jobs[j];
    for (int k : job.part[x]) {
      const Ready& rd = job.rows[k];
      const float* row = job.out + (size_t)k * VAPX_OUT_STRIDE;
      Slot& s = g->slots[rd.slot];
      if (row[VAPX_OUT_STATUS] == 0.f) {
        // u32 len | f64 t | u32 n | x1 | u32 n | x2 | tail   (util.py:122-143; length prefix vap_main.py:446-448)
        uint8_t head[16], mid[4];
        const int tb = encode_tail(g->mode, row, tail.data());
        const uint32_t plen = 8 + 2 * (4 + 8 * (uint32_t)hop) + (uint32_t)tb;
        uint8_t* p = head;
        put_u32(p, plen); put_f64(p, job.t_unix); put_u32(p, (uint32_t)hop);
        p = mid; put_u32(p, (uint32_t)hop);
        double* e = g->f64buf(rd.slot, rd.buf);
        iovec iov[5] = {{head, 16}, {e, (size_t)hop * 8}, {mid, 4}, {e + hop, (size_t)hop * 8}, {tail.data(), (size_t)tb}};
        const size_t total = 4 + plen;
        auto send_to = [&](std::vector<int>& fds) {
          for (size_t i = 0; i < fds.size();) {
            const double ts0 = g->debug ? mono_now() : 0.0;
            const bool sent = send_packet(fds[i], iov, 5, total);
            if (g->debug) atomic_max(g->dbg_send_us, (int64_t)((mono_now() - ts0) * 1e6));   // longest sendmsg() call (the socket is non-blocking)
            if (sent) { g->tx_bytes.fetch_add((int64_t)total, std::memory_order_relaxed); ++i; }
            else { close(fds[i]); fds.erase(fds.begin() + i); g->dropped.fetch_add(1); g->out_conns.fetch_sub(1); if (g->broadcast) link_event(g, LK_OUT_CLOSED, -1); }
          }
        };
        if (g->broadcast) { std::lock_guard<std::mutex> lk(g->slots_mu); send_to(g->out_all); }
        else {
          size_t gone = 0;
          {
            std::lock_guard<std::mutex> lk(s.lmu);
            const size_t before = s.listeners.size();
            if (before == 0) s.dbg_nolistener.fetch_add(1, std::memory_order_relaxed);
            else if (s.dbg_sent.fetch_add(1, std::memory_order_relaxed) == 0) s.dbg_t_first_sent = mono_now();
            send_to(s.listeners);
            gone = before - s.listeners.size();
          }
          for (size_t d = 0; d < gone; ++d) listener_dropped(g, rd.slot);   // (slots_mu is never taken under a slot's lmu here)
        }
        const double lat = mono_now() - rd.t;
        g->lat.add(lat);
        if (lat > 0.010) g->late10.fetch_add(1, std::memory_order_relaxed);
      }
      release_buf(g, rd.slot, rd.buf);
    }
    ++next;
    {
      std::lock_guard<std::mutex> lk(g->job_mu);
      if (++job.senders_done >= g->X) {      // every packet of this tick is out: the Job may be reused
        if (g->debug) atomic_max(g->dbg_job_tx_us, (int64_t)((unix_now() - job.t_unix) * 1e6));   // results ready -> last packet of the tick handed to the kernel
        job.busy = false;
        g->job_done_cv.notify_all();
      }
    }
  }
}

void tick_main(vapx_ingest* g) {
  std::deque<Ready> pending;
  std::vector<uint8_t> in_batch(g->S, 0);
  double first_ready = 0.0;
  int jsel = 0;
  const double max_wait = (g->cfg.max_wait_us > 0 ? g->cfg.max_wait_us : 2000) * 1e-6;
  const double util = (g->cfg.target_util_pct > 0 ? std::min(g->cfg.target_util_pct, 100) : 90) * 0.01;   // 90: measured best at the edge (6144 streams: client p99 12.6 ms vs 19.0 at 75, 16.0 unpaced)
  double earliest_next = 0.0;              // pacing: previous tick's start + its duration / util
  while (!g->stop.load()) {
    std::vector<Ready> fresh;
    std::vector<std::pair<int, int>> resets;
    {
      std::unique_lock<std::mutex> lk(g->ready_mu);
      if (g->ready.empty() && g->resets.empty()) {
        if (pending.empty()) cv_wait_for(g->ready_cv, lk, std::chrono::milliseconds(20));
        else {
          const double left = first_ready + max_wait - mono_now();
          if (left > 0) cv_wait_for(g->ready_cv, lk, std::chrono::microseconds((long)(left * 1e6) + 1));
        }
      }
      fresh.swap(g->ready);
      resets.swap(g->resets);
    }
    for (auto& rs : resets)
      if (g->reset) g->reset(g->user, rs.second ? -(rs.first + 1) : rs.first);   // negative = carry only
    for (auto& rd : fresh) {
      if (rd.gen != g->slots[rd.slot].gen.load(std::memory_order_acquire)) { release_buf(g, rd.slot, rd.buf); continue; }   // connection already gone
      if (pending.empty()) first_ready = rd.t;
      pending.push_back(rd);
    }
    if (pending.empty()) continue;
    const int connected = (int)g->in_conns.load();
    int want = g->cfg.min_batch > 0 ? g->cfg.min_batch : connected;
    want = std::max(1, std::min(want, std::min(connected > 0 ? connected : 1, g->max_batch)));
    const double now = mono_now();
    if ((int)pending.size() < want && now - first_ready < max_wait) continue;
    // pacing: let the batch grow instead of the queue — but never hold a frame longer than max_wait for it (under overload
    // the oldest frame is already older than that and ticks run back to back)
    if (now < earliest_next && (int)pending.size() < g->max_batch && now - first_ready < max_wait) {
      std::unique_lock<std::mutex> lk(g->ready_mu);
      if (g->ready.empty()) cv_wait_for(g->ready_cv, lk, std::chrono::microseconds((long)((earliest_next - now) * 1e6) + 1));
      continue;
    }

    Job& job = g->jobs[jsel];
    {
      std::unique_lock<std::mutex> lk(g->job_mu);
      g->job_done_cv.wait(lk, [&] { return !job.busy || g->stop.load(); });
      if (g->stop.load()) break;
    }
    // one frame per stream per tick, oldest first; a second frame of the same stream waits for the next tick
    job.rows.clear();
    std::deque<Ready> later;
    while (!pending.empty()) {
      Ready rd = pending.front();
      pending.pop_front();
      if (rd.gen != g->slots[rd.slot].gen.load(std::memory_order_acquire)) { release_buf(g, rd.slot, rd.buf); continue; }
      if (in_batch[rd.slot] || (int)job.rows.size() >= g->max_batch) { later.push_back(rd); continue; }
      in_batch[rd.slot] = 1;
      g->slots[rd.slot].state[rd.buf].store(B_INFLIGHT);
      job.rows.push_back(rd);
    }
    pending.swap(later);
    if (!pending.empty()) first_ready = pending.front().t;
    const int n = (int)job.rows.size();
    if (n == 0) continue;
    const size_t fb = (size_t)2 * g->hop * sizeof(float);
    for (int k = 0; k < n; ++k) {
      in_batch[job.rows[k].slot] = 0;
      g->batch_ids[k] = job.rows[k].slot;
      memcpy(g->batch_audio + (size_t)k * 2 * g->hop, g->f32buf(job.rows[k].slot, job.rows[k].buf), fb);
    }
    const double t0 = mono_now();
    const int rc = g->step(g->user, n, g->batch_ids.data(), g->batch_audio, job.out);
    const double t1 = mono_now();
    earliest_next = t0 + (t1 - t0) / util;
    g->step_us.fetch_add((int64_t)((t1 - t0) * 1e6));
    if (g->debug) atomic_max(g->dbg_step_us, (int64_t)((t1 - t0) * 1e6));
    g->ticks.fetch_add(1);
    g->batch_sum.fetch_add(n);
    if (rc != 0 && rc != VAPX_E_NUMERIC) {   // the step itself failed: nothing to send; free the frames and keep serving
      char buf[96];
      snprintf(buf, sizeof buf, "step failed with code %d", rc);
      g->err = buf;
      for (int k = 0; k < n; ++k) release_buf(g, job.rows[k].slot, job.rows[k].buf);
      continue;
    }
    for (int k = 0; k < n; ++k)
      if (job.out[(size_t)k * VAPX_OUT_STRIDE + VAPX_OUT_STATUS] != 0.f) {   // poisoned stream: fresh state, no packet
        if (g->reset) g->reset(g->user, job.rows[k].slot);
        g->numeric_resets.fetch_add(1);
      }
    g->frames_done.fetch_add(n);
    job.t_unix = unix_now();
    for (auto& pt : job.part) pt.clear();
    for (int k = 0; k < n; ++k) job.part[job.rows[k].slot % g->X].push_back(k);
    {
      std::lock_guard<std::mutex> lk(g->job_mu);
      job.n = n;
      job.senders_done = 0;
      job.busy = true;
      ++g->jobs_published;                   // this job is number jobs_published - 1 and sits in jobs[jsel]: publications alternate slots
    }
    g->job_cv.notify_all();
    jsel ^= 1;
  }
}

int engine_step(void* user, int32_t n, const int32_t* ids, const float* audio, float* out) {
  vapx_ingest* g = (vapx_ingest*)user;
  return vapx_step(g->engine, n, ids, audio, g->hop, out, VAPX_AUDIO_HOST | VAPX_OUT_HOST, nullptr);
}
void engine_reset(void* user, int32_t sid) {
  vapx_ingest* g = (vapx_ingest*)user;
  if (sid < 0) (void)vapx_reset_carry(g->engine, -sid - 1);
  else (void)vapx_reset_stream(g->engine, sid);
}

// pin a thread to one core of the configured range (vapx_ingest_config.cpu_first / cpu_count); failures are ignored: placement is a
// tail-latency measure, not a correctness one
void pin_to(std::thread& t, const vapx_ingest_config& c, int k) {
  if (c.cpu_count <= 0 || !t.joinable()) return;
  cpu_set_t set;
  CPU_ZERO(&set);
  if (c.flags & VAPX_INGEST_CORE_SET) {
    for (int i = 0; i < c.cpu_count; ++i) CPU_SET((c.cpu_first + i) % CPU_SETSIZE, &set);
  } else {
    CPU_SET((c.cpu_first + (k % c.cpu_count)) % CPU_SETSIZE, &set);
  }
  (void)pthread_setaffinity_np(t.native_handle(), sizeof set, &set);
}

// the caller's config, whatever its vintage: the struct only ever grows at the end and says how long it is
bool read_config(const vapx_ingest_config* cfg, vapx_ingest_config* out) {
  constexpr int32_t kFirst = (int32_t)offsetof(vapx_ingest_config, cpu_first);   // ABI 2 as first shipped: up to `reserved` (now `flags`)
  if (!cfg || (cfg->struct_size != kFirst && cfg->struct_size != (int32_t)sizeof(vapx_ingest_config))) return false;
  memset(out, 0, sizeof *out);
  memcpy(out, cfg, (size_t)cfg->struct_size);
  out->struct_size = (int32_t)sizeof(vapx_ingest_config);
  return true;
}

int open_common(vapx_ingest* g, const vapx_ingest_config* cfg_in) {
  if (!read_config(cfg_in, &g->cfg)) return VAPX_E_INVAL;
  const vapx_ingest_config* cfg = &g->cfg;
  g->debug = getenv("VAPX_INGEST_DEBUG") != nullptr;
  g->no_lowat = getenv("VAPX_INGEST_NO_LOWAT") != nullptr;
  if (g->cfg.gain == 0.0) g->cfg.gain = 1.0;
  g->R = cfg->rx_threads > 0 ? std::min(cfg->rx_threads, 16) : 2;
  g->X = cfg->tx_threads > 0 ? std::min(cfg->tx_threads, 16) : 2;
  g->broadcast = cfg->broadcast < 0 ? (g->S == 1) : (cfg->broadcast != 0);
  g->hop = 16000 / g->hz;
  g->slots.reset(new Slot[g->S]);
  const size_t per = (size_t)g->S * NBUF * 2 * g->hop;
  const size_t ba = (size_t)g->max_batch * 2 * g->hop;
  const size_t ob = (size_t)g->max_batch * VAPX_OUT_STRIDE;
  // page-locked staging so vapx_step DMAs straight out of / into it; without a HIP device (host-logic tests over a step
  // function) plain memory does
  g->stage = (float*)vapx_host_alloc(per * sizeof(float));
  g->pinned_blocks = g->stage != nullptr;
  auto grab = [&](size_t n) { return (float*)(g->pinned_blocks ? vapx_host_alloc(n * sizeof(float)) : calloc(n, sizeof(float))); };
  if (!g->stage) g->stage = grab(per);
  g->echo.assign(per, 0.0);
  g->batch_audio = grab(ba);
  g->batch_ids.assign(g->max_batch, 0);
  for (auto& j : g->jobs) {
    j.out = grab(ob);
    if (j.out) memset(j.out, 0, ob * sizeof(float));   // hipHostMalloc memory is not zeroed: a step function may leave columns unset
    j.rows.reserve(g->max_batch);
    j.part.assign(g->X, {});
  }
  if (!g->stage || !g->batch_audio || !g->jobs[0].out || !g->jobs[1].out) return VAPX_E_NOMEM;
  // One descriptor per input connection and one per listener: 2 x S and some.  Raise the soft RLIMIT_NOFILE towards the hard limit if it
  // is short, and say so if the hard limit is short too: connections beyond it wait in the listen queue (accept4 fails with EMFILE, counted
  // in accept_fd_errors), and a dialogue whose output connection is not attached yet loses its results — they go to nobody
  if (!(cfg->flags & VAPX_INGEST_KEEP_NOFILE)) {
    const rlim_t need = (rlim_t)2 * (rlim_t)g->S + 256;
    rlimit rl;
    if (getrlimit(RLIMIT_NOFILE, &rl) == 0 && rl.rlim_cur != RLIM_INFINITY && rl.rlim_cur < need) {
      rlimit want = rl;
      want.rlim_cur = rl.rlim_max == RLIM_INFINITY ? need : std::min<rlim_t>(rl.rlim_max, std::max<rlim_t>(need, rl.rlim_cur));
      (void)setrlimit(RLIMIT_NOFILE, &want);
      if (getrlimit(RLIMIT_NOFILE, &rl) == 0 && rl.rlim_cur < need)
        fprintf(stderr, "[vapx ingest] RLIMIT_NOFILE is %llu (hard limit), %llu are needed for %d dialogue slots with one listener each: later "
                "connections will wait in the listen queue\n", (unsigned long long)rl.rlim_cur, (unsigned long long)need, g->S);
    }
  }
  // port_in < 0: a PASSIVE shard of a multi-GPU front door (vapx_frontdoor_open): it listens on nothing, connections are handed to it
  const bool passive = cfg->port_in < 0;
  if (passive != (cfg->port_out < 0)) return VAPX_E_INVAL;   // a shard is passive on BOTH ports or on none (-1 / >= 0 mixed is a configuration error)
  if (!passive) {
    g->lin = listen_on(cfg->port_in, cfg->bind_any != 0, &g->port_in);
    g->lout = listen_on(cfg->port_out, cfg->bind_any != 0, &g->port_out);
    if (g->lin < 0 || g->lout < 0) return VAPX_E_INVAL;
  }
  g->resume_mu = std::vector<std::mutex>(g->R);
  g->resume.assign(g->R, {});
  for (int r = 0; r < g->R; ++r) {
    g->ep.push_back(epoll_create1(0));
    g->wake.push_back(eventfd(0, EFD_NONBLOCK));
    ep_add(g->ep[r], g->wake[r], K_WAKE);
  }
  g->lcount.assign(g->S, 0);
  if (!passive) {
    g->ep_accept = epoll_create1(0);          // own thread: a connect storm must not starve the streams of a receive thread
    ep_add(g->ep_accept, g->lin, K_LISTEN_IN);
    ep_add(g->ep_accept, g->lout, K_LISTEN_OUT);
    g->accept_thread = std::thread(accept_main, g);
  }
  for (int r = 0; r < g->R; ++r) g->rx_threads.emplace_back(rx_main, g, r);
  for (int x = 0; x < g->X; ++x) g->tx_threads.emplace_back(tx_main, g, x);
  g->tick_thread = std::thread(tick_main, g);
  // placement: tick + accept on the first core of the range, then one core per receive thread, then one per sender
  pin_to(g->tick_thread, g->cfg, 0);
  pin_to(g->accept_thread, g->cfg, 0);
  for (int r = 0; r < g->R; ++r) pin_to(g->rx_threads[r], g->cfg, 1 + r);
  for (int x = 0; x < g->X; ++x) pin_to(g->tx_threads[x], g->cfg, 1 + g->R + x);
  return VAPX_OK;
}

}  // namespace

extern "C" {

int vapx_ingest_open_fn(vapx_ingest_step_fn step, vapx_ingest_reset_fn reset, void* user, int32_t n_streams, int32_t max_batch,
                        int32_t frame_hz, int32_t mode, const vapx_ingest_config* cfg, vapx_ingest_handle* out) {
  vapx_ingest_config probe;
  if (!step || !out || !read_config(cfg, &probe)) return VAPX_E_INVAL;
  if (n_streams < 1 || max_batch < 1 || max_batch > n_streams) return VAPX_E_INVAL;
  if (frame_hz != 5 && frame_hz != 10 && frame_hz != 20 && frame_hz != 50) return VAPX_E_INVAL;
  if (mode < 0 || mode > 2) return VAPX_E_INVAL;
  vapx_ingest* g = new vapx_ingest();
  g->step = step; g->reset = reset; g->user = user;
  g->S = n_streams; g->max_batch = max_batch; g->hz = frame_hz; g->mode = mode;
  int rc = open_common(g, cfg);
  if (rc != VAPX_OK) { vapx_ingest_close(g); return rc; }
  *out = g;
  return VAPX_OK;
}

int vapx_ingest_open(vapx_handle engine, const vapx_ingest_config* cfg, vapx_ingest_handle* out) {
  vapx_ingest_config probe;
  if (!engine || !out || !read_config(cfg, &probe)) return VAPX_E_INVAL;
  vapx_config ec;
  int rc = vapx_get_config(engine, &ec);
  if (rc != VAPX_OK) return rc;
  vapx_ingest* g = new vapx_ingest();
  g->engine = engine;
  g->step = engine_step; g->reset = engine_reset; g->user = g;
  g->S = ec.max_streams; g->max_batch = ec.max_batch; g->hz = ec.frame_hz; g->mode = ec.mode;
  {  // warm the engine up before the first client connects: the first vapx_step of a process loads the code objects and sizes
     // the runtime's pools (hundreds of ms) — paid here on silence, then every touched stream is reset
    const int nw = ec.max_batch, hop = 16000 / ec.frame_hz;
    float* a = (float*)vapx_host_alloc((size_t)nw * 2 * hop * sizeof(float));
    float* o = (float*)vapx_host_alloc((size_t)nw * VAPX_OUT_STRIDE * sizeof(float));
    if (a && o) {
      memset(a, 0, (size_t)nw * 2 * hop * sizeof(float));
      for (int k = 0; k < 3; ++k) (void)vapx_step(engine, nw, nullptr, a, hop, o, VAPX_AUDIO_HOST | VAPX_OUT_HOST, nullptr);
      for (int i = 0; i < nw; ++i) (void)vapx_reset_stream(engine, i);
    }
    vapx_host_free(a);
    vapx_host_free(o);
  }
  rc = open_common(g, cfg);
  if (rc != VAPX_OK) { vapx_ingest_close(g); return rc; }
  *out = g;
  return VAPX_OK;
}

int vapx_ingest_ports(vapx_ingest_handle g, int32_t* port_in, int32_t* port_out) {
  if (!g) return VAPX_E_INVAL;
  if (port_in) *port_in = g->port_in;
  if (port_out) *port_out = g->port_out;
  return VAPX_OK;
}

int vapx_ingest_stats_read(vapx_ingest_handle g, vapx_ingest_stats* st, int32_t reset_latency_window) {
  if (!g || !st) return VAPX_E_INVAL;
  memset(st, 0, sizeof *st);
  st->frames_done = g->frames_done.load();
  st->ticks = g->ticks.load();
  st->rx_bytes = g->rx_bytes.load();
  st->tx_bytes = g->tx_bytes.load();
  st->in_connections = g->in_conns.load();
  st->out_connections = g->out_conns.load();
  st->dropped_listeners = g->dropped.load();
  st->numeric_resets = g->numeric_resets.load();
  st->overruns = g->overruns.load();
  st->mean_batch = st->ticks ? (double)g->batch_sum.load() / (double)st->ticks : 0.0;
  const int64_t c = g->lat.cnt.load();
  st->lat_mean_ms = c ? (double)g->lat.sum_us.load() / (double)c * 1e-3 : 0.0;
  st->lat_p50_ms = g->lat.pct(0.50);
  st->lat_p99_ms = g->lat.pct(0.99);
  st->lat_max_ms = (double)g->lat.max_us.load() * 1e-3;
  st->step_mean_ms = st->ticks ? (double)g->step_us.load() / (double)st->ticks * 1e-3 : 0.0;
  if (reset_latency_window) { g->lat.clear(); g->late10.store(0); }
  return VAPX_OK;
}

int vapx_ingest_late_read(vapx_ingest_handle g, int64_t* over_10ms, int64_t* answered) {
  if (!g) return VAPX_E_INVAL;
  if (over_10ms) *over_10ms = g->late10.load();
  if (answered) *answered = g->lat.cnt.load();
  return VAPX_OK;
}

void vapx_ingest_close(vapx_ingest_handle g) {
  if (!g) return;
  g->stop.store(true);
  g->ready_cv.notify_all();
  g->job_cv.notify_all();
  g->job_done_cv.notify_all();
  for (int fd : g->wake) kick(fd);
  if (g->accept_thread.joinable()) g->accept_thread.join();
  if (g->link_thread.joinable()) g->link_thread.join();
  if (g->link_fd >= 0) { std::lock_guard<std::mutex> lk(g->link_mu); g->link_fd = -1; }   // (the descriptor stays the caller's: include/vapx.h)
  if (g->ep_accept >= 0) close(g->ep_accept);
  for (auto& t : g->rx_threads) if (t.joinable()) t.join();
  if (g->tick_thread.joinable()) g->tick_thread.join();
  g->job_cv.notify_all();
  for (auto& t : g->tx_threads) if (t.joinable()) t.join();
  if (g->debug)
    fprintf(stderr, "[vapx ingest] longest rx pass %.2f ms, longest gap between busy rx passes %.2f ms, longest step %.2f ms, latency max %.2f ms, "
            "longest recv() %.2f ms, longest sendmsg() %.2f ms, longest tick fan-out (results ready -> last packet sent) %.2f ms\n",
            g->dbg_rx_pass_us.load() * 1e-3, g->dbg_rx_gap_us.load() * 1e-3, g->dbg_step_us.load() * 1e-3, (double)g->lat.max_us.load() * 1e-3,
            g->dbg_recv_us.load() * 1e-3, g->dbg_send_us.load() * 1e-3, g->dbg_job_tx_us.load() * 1e-3);
  if (g->debug && g->slots) {
    // streams whose results went nowhere (no output connection attached yet) or whose frames did not all come back out
    int shown = 0, odd = 0;
    for (int i = 0; i < g->S; ++i) {
      Slot& s = g->slots[i];
      const int64_t rdy = s.dbg_ready.load(), snt = s.dbg_sent.load(), nol = s.dbg_nolistener.load();
      if (nol == 0 && rdy == snt) continue;
      ++odd;
      if (shown++ < 24)
        fprintf(stderr, "[vapx ingest] slot %d: %ld frames assembled, %ld sent, %ld had NO listener; first frame ready %.3f s, listener attached %.3f s, first packet sent %.3f s (after the first slot's first frame)\n",
                i, (long)rdy, (long)snt, (long)nol, s.dbg_t_first_ready - g->slots[0].dbg_t_first_ready, s.dbg_t_attach - g->slots[0].dbg_t_first_ready,
                s.dbg_t_first_sent - g->slots[0].dbg_t_first_ready);
    }
    fprintf(stderr, "[vapx ingest] accept4 failures for lack of descriptors: %ld; CLOCK_MONOTONIC: inputs adopted %.3f .. %.3f, outputs adopted %.3f .. %.3f, slot 0's first frame %.3f\n",
            (long)g->accept_fd_errors.load(), g->dbg_t_in_first, g->dbg_t_in_last, g->dbg_t_out_first, g->dbg_t_out_last, g->slots[0].dbg_t_first_ready);
    fprintf(stderr, "[vapx ingest] %d of %d slots lost results to a missing listener or did not send every assembled frame\n", odd, g->S);
  }
  if (g->slots) {
    for (int i = 0; i < g->S; ++i) {
      if (g->slots[i].fd_in >= 0) close(g->slots[i].fd_in);
      for (int fd : g->slots[i].listeners) close(fd);
    }
  }
  for (int fd : g->out_all) close(fd);
  if (g->lin >= 0) close(g->lin);
  if (g->lout >= 0) close(g->lout);
  for (int fd : g->ep) close(fd);
  for (int fd : g->wake) close(fd);
  auto drop = [&](void* p) {
    if (!p) return;
    if (g->pinned_blocks) vapx_host_free(p); else free(p);
  };
  drop(g->stage);
  drop(g->batch_audio);
  drop(g->jobs[0].out);
  drop(g->jobs[1].out);
  delete g;
}

// ---- one front door for N per-GPU front-ends ------------------------------------------------------------------------------------
// The reference listens on ONE port pair (vap_main.py:338-366).  N GPUs = N passive shards (each with its own engine, receive / tick /
// send threads) behind one accept thread on that pair.  Dialogue slots are numbered GLOBALLY g = local_slot * N + shard, so that
//   - a new input connection takes the lowest free global slot: GPUs fill evenly at any load (round-robin), and a dialogue that
//     reconnects while its slot is still the lowest free one lands on the GPU that holds its state (ring, LSTM) — stickiness;
//   - the k-th output connection hears the k-th dialogue, exactly as a single front-end does (fewest listeners, lowest global slot).
// A dialogue never moves between GPUs: its state lives there (vapx_get_state / vapx_set_state migrate one on purpose).
}  // extern "C"

struct RemoteShard {                        // the door's mirror of a shard that lives in another process
  int link = -1, S = 0, hz = 0, mode = 0;
  bool broadcast = false, dead = false;
  std::vector<char> used;                   // input slot taken
  int free_hint = 0;
  std::vector<int> lcount;                  // listeners per slot
  int lmin = 0, lcursor = 0, bcast = 0;
  int lowest_free() {
    for (int i = free_hint; i < S; ++i) if (!used[i]) { free_hint = i; return i; }
    free_hint = S;
    return -1;
  }
  std::pair<int, int> next_listener() {
    if (broadcast) return {bcast, 0};
    for (int pass = 0; pass < 2; ++pass) {
      for (int i = lcursor; i < S; ++i) if (lcount[i] == lmin) { lcursor = i; return {lmin, i}; }
      ++lmin; lcursor = 0;
    }
    return {lmin, 0};
  }
};

struct vapx_frontdoor {
  std::vector<vapx_ingest*> shards;
  std::vector<RemoteShard> remote;          // vapx_frontdoor_open_links: shards in other processes
  int lin = -1, lout = -1, port_in = 0, port_out = 0, ep = -1;
  std::thread th;
  std::atomic<bool> stop{false};
  std::atomic<int64_t> accepted_in{0}, accepted_out{0}, refused{0};
};

namespace {

void frontdoor_main(vapx_frontdoor* d) {
  epoll_event evs[8];
  const int N = (int)d->shards.size();
  while (!d->stop.load()) {
    int n = epoll_wait(d->ep, evs, 8, 100);
    for (int i = 0; i < n; ++i) {
      const uint64_t kind = evs[i].data.u64 & ~0xffffffffull;
      for (;;) {
        int fd = accept4(kind == K_LISTEN_IN ? d->lin : d->lout, nullptr, nullptr, SOCK_NONBLOCK);
        if (fd < 0) {
          // out of descriptors: the listen socket stays readable (level-triggered), so epoll_wait would return at once and this thread
          // would spin at 100 % — back off until a connection closes
          if (errno == EMFILE || errno == ENFILE || errno == ENOBUFS || errno == ENOMEM) std::this_thread::sleep_for(std::chrono::milliseconds(20));
          break;
        }
        if (kind == K_LISTEN_IN) {
          bool placed = false;
          for (int attempt = 0; attempt < N && !placed; ++attempt) {      // (a shard can fill up between the query and the adoption)
            long best = -1; int who = -1;
            for (int k = 0; k < N; ++k) {
              const int ls = lowest_free_slot(d->shards[k]);
              if (ls < 0) continue;
              const long gslot = (long)ls * N + k;
              if (who < 0 || gslot < best) { best = gslot; who = k; }
            }
            if (who < 0) break;
            placed = adopt_in(d->shards[who], fd);
          }
          if (placed) d->accepted_in.fetch_add(1);
          else { close(fd); d->refused.fetch_add(1); }               // every dialogue slot of every GPU is taken
        } else {
          int who = 0; long bestc = -1, bestg = -1;
          for (int k = 0; k < N; ++k) {
            // a broadcast shard (one dialogue slot: serve.py's default --streams 1) is ONE candidate: (its listener count, slot 0) — so the
            // k-th output connection hears GPU k mod N's dialogue instead of every listener piling up on shard 0
            std::pair<int, int> c;
            if (d->shards[k]->broadcast) { std::lock_guard<std::mutex> lk(d->shards[k]->slots_mu); c = {(int)d->shards[k]->out_all.size(), 0}; }
            else c = next_listener_slot(d->shards[k]);
            const long gslot = (long)c.second * N + k;
            if (bestc < 0 || c.first < bestc || (c.first == bestc && gslot < bestg)) { bestc = c.first; bestg = gslot; who = k; }
          }
          adopt_out(d->shards[who], fd);
          d->accepted_out.fetch_add(1);
        }
      }
    }
  }
}

constexpr uint64_t K_LINK = 4ull << 32;

// The same door with the shards in OTHER processes (one per GPU): placement runs on the mirrors, the accepted socket travels to the worker
// (SCM_RIGHTS) and is closed here - this process never holds more than the listen sockets, the links and the connection in hand.
void frontdoor_remote_main(vapx_frontdoor* d) {
  epoll_event evs[16];
  const int N = (int)d->remote.size();
  while (!d->stop.load()) {
    int n = epoll_wait(d->ep, evs, 16, 100);
    for (int i = 0; i < n; ++i) {
      const uint64_t kind = evs[i].data.u64 & ~0xffffffffull;
      if (kind == K_LINK) {                  // a worker reports a released slot / a dropped listener (or went away)
        RemoteShard& r = d->remote[(int)(evs[i].data.u64 & 0xffffffffu)];
        for (;;) {
          LinkMsg m;
          const int rc = link_recv(r.link, &m, nullptr, 0);
          if (rc == 0) break;
          if (rc < 0) { if (!r.dead) { r.dead = true; epoll_ctl(d->ep, EPOLL_CTL_DEL, r.link, nullptr); } break; }
          if (m.kind == LK_IN_CLOSED && m.a >= 0 && m.a < r.S) { r.used[m.a] = 0; if (m.a < r.free_hint) r.free_hint = m.a; }
          else if (m.kind == LK_OUT_CLOSED) {
            if (r.broadcast || m.a < 0) { if (r.bcast > 0) --r.bcast; }
            else if (m.a < r.S && r.lcount[m.a] > 0) {
              if (--r.lcount[m.a] < r.lmin) { r.lmin = r.lcount[m.a]; r.lcursor = m.a; }
              else if (r.lcount[m.a] == r.lmin && m.a < r.lcursor) r.lcursor = m.a;
            }
          }
        }
        continue;
      }
      for (;;) {
        int fd = accept4(kind == K_LISTEN_IN ? d->lin : d->lout, nullptr, nullptr, SOCK_NONBLOCK);
        if (fd < 0) {
          if (errno == EMFILE || errno == ENFILE || errno == ENOBUFS || errno == ENOMEM) std::this_thread::sleep_for(std::chrono::milliseconds(20));
          break;
        }
        if (kind == K_LISTEN_IN) {
          long best = -1; int who = -1, ls_who = -1;
          for (int k = 0; k < N; ++k) {
            if (d->remote[k].dead) continue;
            const int ls = d->remote[k].lowest_free();
            if (ls < 0) continue;
            const long gslot = (long)ls * N + k;
            if (who < 0 || gslot < best) { best = gslot; who = k; ls_who = ls; }
          }
          if (who >= 0 && link_send(d->remote[who].link, nullptr, LinkMsg{LK_ADOPT_IN, ls_who, 0, 0}, fd)) {
            d->remote[who].used[ls_who] = 1;
            d->accepted_in.fetch_add(1);
          } else d->refused.fetch_add(1);      // every dialogue slot of every GPU is taken (or that worker is gone)
        } else {
          int who = -1, slot = 0; long bestc = -1, bestg = -1;
          for (int k = 0; k < N; ++k) {
            if (d->remote[k].dead) continue;
            const std::pair<int, int> c = d->remote[k].next_listener();
            const long gslot = (long)c.second * N + k;
            if (bestc < 0 || c.first < bestc || (c.first == bestc && gslot < bestg)) { bestc = c.first; bestg = gslot; who = k; slot = c.second; }
          }
          if (who >= 0 && link_send(d->remote[who].link, nullptr, LinkMsg{LK_ADOPT_OUT, slot, 0, 0}, fd)) {
            RemoteShard& r = d->remote[who];
            if (r.broadcast) ++r.bcast; else { ++r.lcount[slot]; r.lcursor = slot + 1; }
            d->accepted_out.fetch_add(1);
          } else d->refused.fetch_add(1);
        }
        close(fd);                           // the worker holds its own copy now
      }
    }
  }
}

}  // namespace

extern "C" {

int vapx_ingest_attach_link(vapx_ingest_handle g, int32_t link_fd) {
  if (!g || link_fd < 0 || g->link_fd >= 0) return VAPX_E_INVAL;
  if (g->lin >= 0 || g->lout >= 0) return VAPX_E_INVAL;           // only a passive shard takes its connections from a door
  g->link_fd = link_fd;
  if (!link_send(link_fd, &g->link_mu, LinkMsg{LK_HELLO, g->S, g->hz, g->mode | (g->broadcast ? 256 : 0)})) { g->link_fd = -1; return VAPX_E_INVAL; }
  g->link_thread = std::thread(link_main, g);
  return VAPX_OK;
}

int vapx_frontdoor_open_links(const int32_t* link_fds, int32_t n_links, int32_t port_in, int32_t port_out, int32_t bind_any, vapx_frontdoor_handle* out) {
  if (!link_fds || n_links < 1 || !out || port_in < 0 || port_out < 0) return VAPX_E_INVAL;
  vapx_frontdoor* d = new vapx_frontdoor();
  d->remote.resize(n_links);
  for (int k = 0; k < n_links; ++k) {        // every worker introduces itself (it may still be loading its weights: wait up to 5 minutes)
    RemoteShard& r = d->remote[k];
    r.link = link_fds[k];
    LinkMsg m;
    int rc = 0;
    for (int waited = 0; waited < 3000 && rc == 0; ++waited) rc = link_recv(r.link, &m, nullptr, 100);
    if (rc <= 0 || m.kind != LK_HELLO || m.a < 1) { delete d; return VAPX_E_INVAL; }
    r.S = m.a; r.hz = m.b; r.mode = m.c & 255; r.broadcast = (m.c & 256) != 0;
    r.used.assign(r.S, 0); r.lcount.assign(r.S, 0);
    if (r.hz != d->remote[0].hz || r.mode != d->remote[0].mode) { delete d; return VAPX_E_INVAL; }
  }
  d->lin = listen_on(port_in, bind_any != 0, &d->port_in);
  d->lout = listen_on(port_out, bind_any != 0, &d->port_out);
  if (d->lin < 0 || d->lout < 0) { vapx_frontdoor_close(d); return VAPX_E_INVAL; }
  d->ep = epoll_create1(0);
  ep_add(d->ep, d->lin, K_LISTEN_IN);
  ep_add(d->ep, d->lout, K_LISTEN_OUT);
  for (int k = 0; k < n_links; ++k) ep_add(d->ep, d->remote[k].link, K_LINK | (uint32_t)k);
  d->th = std::thread(frontdoor_remote_main, d);
  *out = d;
  return VAPX_OK;
}

int vapx_frontdoor_open(vapx_ingest_handle* shards, int32_t n_shards, int32_t port_in, int32_t port_out, int32_t bind_any,
                        vapx_frontdoor_handle* out) {
  if (!shards || n_shards < 1 || !out || port_in < 0 || port_out < 0) return VAPX_E_INVAL;
  for (int k = 0; k < n_shards; ++k) {
    if (!shards[k] || shards[k]->lin >= 0 || shards[k]->lout >= 0) return VAPX_E_INVAL;   // shards must be passive (port_in = port_out = -1)
    if (shards[k]->hz != shards[0]->hz || shards[k]->mode != shards[0]->mode) return VAPX_E_INVAL;
  }
  vapx_frontdoor* d = new vapx_frontdoor();
  d->shards.assign(shards, shards + n_shards);
  d->lin = listen_on(port_in, bind_any != 0, &d->port_in);
  d->lout = listen_on(port_out, bind_any != 0, &d->port_out);
  if (d->lin < 0 || d->lout < 0) { vapx_frontdoor_close(d); return VAPX_E_INVAL; }
  d->ep = epoll_create1(0);
  ep_add(d->ep, d->lin, K_LISTEN_IN);
  ep_add(d->ep, d->lout, K_LISTEN_OUT);
  d->th = std::thread(frontdoor_main, d);
  *out = d;
  return VAPX_OK;
}

int vapx_frontdoor_ports(vapx_frontdoor_handle d, int32_t* port_in, int32_t* port_out) {
  if (!d) return VAPX_E_INVAL;
  if (port_in) *port_in = d->port_in;
  if (port_out) *port_out = d->port_out;
  return VAPX_OK;
}

int vapx_frontdoor_counts(vapx_frontdoor_handle d, int64_t* accepted_in, int64_t* accepted_out, int64_t* refused) {
  if (!d) return VAPX_E_INVAL;
  if (accepted_in) *accepted_in = d->accepted_in.load();
  if (accepted_out) *accepted_out = d->accepted_out.load();
  if (refused) *refused = d->refused.load();
  return VAPX_OK;
}

void vapx_frontdoor_close(vapx_frontdoor_handle d) {
  if (!d) return;
  d->stop.store(true);
  if (d->th.joinable()) d->th.join();
  if (d->ep >= 0) close(d->ep);
  if (d->lin >= 0) close(d->lin);
  if (d->lout >= 0) close(d->lout);
  delete d;                                  // the shards stay open: close them with vapx_ingest_close
}

int64_t vapx_wire_decode_input(const uint8_t* bytes, size_t n_bytes, double gain, float* x1_f32, float* x2_f32, double* x1_f64,
                               double* x2_f64) {
  if (!bytes || n_bytes % PAIR_BYTES) return VAPX_E_INVAL;   // util.conv_bytearray_2_2floatarray needs whole f64 pairs
  const size_t n = n_bytes / PAIR_BYTES;
  for (size_t i = 0; i < n; ++i) {
    double v[2];
    memcpy(v, bytes + i * PAIR_BYTES, PAIR_BYTES);
    const double a = gain != 1.0 ? v[0] * gain : v[0], b = gain != 1.0 ? v[1] * gain : v[1];
    if (x1_f64) x1_f64[i] = a;
    if (x2_f64) x2_f64[i] = b;
    if (x1_f32) x1_f32[i] = (float)a;
    if (x2_f32) x2_f32[i] = (float)b;
  }
  return (int64_t)n;
}

int64_t vapx_wire_encode_result(int32_t mode, double t, const double* x1, const double* x2, int32_t n, const float* row, uint8_t* dst,
                                size_t cap) {
  if (mode < 0 || mode > 2 || !x1 || !x2 || n < 0 || !row) return VAPX_E_INVAL;
  const int nr = mode == VAPX_MODE_NOD ? (int)row[VAPX_OUT_NVALID] : 0;
  if (nr < 0 || nr > 256) return VAPX_E_INVAL;
  const size_t plen = 8 + 2 * (4 + 8 * (size_t)n) + (size_t)tail_bytes(mode, nr);
  if (!dst || cap < 4 + plen) return (int64_t)(4 + plen);
  uint8_t* p = dst;
  put_u32(p, (uint32_t)plen);
  put_f64(p, t);
  put_u32(p, (uint32_t)n);
  memcpy(p, x1, (size_t)n * 8); p += (size_t)n * 8;
  put_u32(p, (uint32_t)n);
  memcpy(p, x2, (size_t)n * 8); p += (size_t)n * 8;
  p += encode_tail(mode, row, p);
  return (int64_t)(p - dst);
}

}  // extern "C"
