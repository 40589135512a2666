// loadgen — real-time load generator for the VAP TCP front-end (reference framing: 2560-byte input packets of 160 x
// {f64 ch1, f64 ch2}; length-prefixed result packets).  Plays S dialogue clients against one server process:
//   * connects S input sockets, then S output sockets (the server pairs the k-th output connection with the k-th stream);
//   * every stream sends its audio in real time — one packet per `--packet-ms` (10 ms like the reference client, or a whole
//     frame at once) — with the streams' frame boundaries spread evenly over the frame period;
//   * receiver threads parse the result packets; latency of a frame = just before its last packet is sent -> complete
//     result packet read.
// Prints one JSON line: frames sent / answered, latency percentiles, late frames (> --late-ms).
// Scale-out (BASELINE config 4: 32768 dialogues behind one port pair; a process here may hold ~20 k descriptors):
//   --inband 1        the send time of a frame and the stream's id ride IN the audio (first sample pair of the frame's last packet, scaled to
//                     ~1e-3 so a real engine is not disturbed); the server echoes the audio in f64, so the receiver reads them back from the
//                     result packet: latency needs no shared state between sender and receiver — any output socket may hear any dialogue, and
//                     several loadgen PROCESSES can drive one server.  Also checks that an output socket keeps hearing the same dialogue.
//   --procs P --rank r --sync-dir D    P processes, rank r plays dialogues r, r + P, ... of --total-streams (phases spread over the GLOBAL
//                     population); all dial their inputs, meet (files in D), dial their outputs, meet, and start at a common CLOCK_MONOTONIC time
//   --src-ips N       dial from 127.0.0.2 .. 127.0.0.(1+N) in turn (one (src ip, dst ip, dst port) triple has ~28 k ephemeral ports)
//   --hist-out F      latency histogram (50 us bins up to 400 ms) as JSON, for merging the processes' percentiles
// Build: make -C vap-realtime_amd/csrc loadgen   (plain C++17, no dependencies)
#include <arpa/inet.h>
#include <errno.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <sys/epoll.h>
#include <sys/prctl.h>
#include <sys/resource.h>
#include <sys/socket.h>
#include <time.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <queue>
#include <string>
#include <thread>
#include <vector>

static double now_s() {
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

static int g_src_ips = 0, g_dial_seq = 0;
static int dial(const char* host, int port) {
  int s = socket(AF_INET, SOCK_STREAM, 0);
  if (g_src_ips > 0) {   // spread the source address over 127.0.0.2.. (the whole 127/8 is local): ports are chosen at connect() time per 4-tuple
    int one = 1;
#ifdef IP_BIND_ADDRESS_NO_PORT
    setsockopt(s, IPPROTO_IP, IP_BIND_ADDRESS_NO_PORT, &one, sizeof one);
#endif
    sockaddr_in b;
    memset(&b, 0, sizeof b);
    b.sin_family = AF_INET;
    b.sin_addr.s_addr = htonl(0x7f000002u + (uint32_t)(g_dial_seq++ % g_src_ips));
    if (bind(s, (sockaddr*)&b, sizeof b) != 0) perror("bind source address");
  }
  sockaddr_in a;
  memset(&a, 0, sizeof a);
  a.sin_family = AF_INET;
  a.sin_port = htons((uint16_t)port);
  inet_pton(AF_INET, host, &a.sin_addr);
  for (int tries = 0; tries < 500; ++tries) {
    if (connect(s, (sockaddr*)&a, sizeof a) == 0) {
      int one = 1;
      setsockopt(s, IPPROTO_TCP, TCP_NODELAY, &one, sizeof one);
      return s;
    }
    usleep(20000);
  }
  perror("connect");
  exit(2);
}

struct Stream {
  int fd_in = -1, fd_out = -1;
  std::mutex mu;
  std::deque<double> sent;          // send-completion time of each frame not yet answered
  std::vector<uint8_t> rbuf;
  long answered = 0, frames = 0;
  int heard = -1;                   // --inband: the dialogue this output socket hears (must never change)
  double t_first_sent = 0, t_first_answer = 0;
};

int main(int argc, char** argv) {
  const char* host = "127.0.0.1";
  int port_in = 50007, port_out = 50008, S = 256, hz = 20, packet_ms = 10, threads = 4;
  double seconds = 10.0, late_ms = 10.0, warm = 3.0;
  int inband = 0, procs = 1, rank = 0, total_streams = 0;
  const char* sync_dir = nullptr;
  const char* hist_out = nullptr;
  for (int i = 1; i + 1 < argc; i += 2) {
    std::string k = argv[i];
    const char* v = argv[i + 1];
    if (k == "--host") host = v;
    else if (k == "--port-in") port_in = atoi(v);
    else if (k == "--port-out") port_out = atoi(v);
    else if (k == "--streams") S = atoi(v);
    else if (k == "--hz") hz = atoi(v);
    else if (k == "--seconds") seconds = atof(v);
    else if (k == "--warm") warm = atof(v);
    else if (k == "--packet-ms") packet_ms = atoi(v);
    else if (k == "--threads") threads = atoi(v);
    else if (k == "--late-ms") late_ms = atof(v);
    else if (k == "--inband") inband = atoi(v);
    else if (k == "--procs") procs = atoi(v);
    else if (k == "--rank") rank = atoi(v);
    else if (k == "--total-streams") total_streams = atoi(v);
    else if (k == "--sync-dir") sync_dir = v;
    else if (k == "--src-ips") g_src_ips = atoi(v);
    else if (k == "--hist-out") hist_out = v;
    else { fprintf(stderr, "unknown option %s\n", k.c_str()); return 2; }
  }
  if (total_streams <= 0) total_streams = S * procs;
  if (procs > 1 && (!sync_dir || !inband)) { fprintf(stderr, "--procs > 1 needs --sync-dir and --inband 1\n"); return 2; }
  // meet the other loadgen processes: touch <dir>/<tag>.<rank>, wait until every rank's file is there
  auto meet = [&](const char* tag) {
    if (procs <= 1) return;
    char p[512];
    snprintf(p, sizeof p, "%s/%s.%d", sync_dir, tag, rank);
    FILE* f = fopen(p, "w"); if (f) fclose(f);
    for (int r = 0; r < procs; ++r) {
      snprintf(p, sizeof p, "%s/%s.%d", sync_dir, tag, r);
      while (access(p, F_OK) != 0) usleep(2000);
    }
  };
  const int hop = 16000 / hz;
  const double period = 1.0 / hz;
  const int packets_per_frame = (int)lround(period * 1000.0 / packet_ms);
  const int pk_samples = hop / packets_per_frame;
  // 8 frames of synthetic two-speaker audio per phase class (harmonic stack + noise), f64 interleaved like the wire
  const int NF = 8;
  std::vector<double> audio((size_t)NF * hop * 2);
  for (int i = 0; i < NF * hop; ++i) {
    const double t = i / 16000.0;
    audio[2 * i] = 0.2 * sin(2 * M_PI * 140.0 * t) * (0.6 + 0.4 * sin(2 * M_PI * 4.0 * t)) + 1e-3 * ((rand() % 2001) / 1000.0 - 1.0);
    audio[2 * i + 1] = 0.15 * sin(2 * M_PI * 210.0 * t + 1.0) * (i / hop % 2 ? 1.0 : 0.05) + 1e-3 * ((rand() % 2001) / 1000.0 - 1.0);
  }
  {   // two sockets per stream: raise the soft descriptor limit if it is short
    rlimit rl;
    const rlim_t need = (rlim_t)2 * S + 256;
    if (getrlimit(RLIMIT_NOFILE, &rl) == 0 && rl.rlim_cur < need) {
      rl.rlim_cur = rl.rlim_max == RLIM_INFINITY ? need : std::min<rlim_t>(rl.rlim_max, need);
      setrlimit(RLIMIT_NOFILE, &rl);
      if (getrlimit(RLIMIT_NOFILE, &rl) == 0 && rl.rlim_cur < need) fprintf(stderr, "loadgen: RLIMIT_NOFILE %llu < %llu needed\n", (unsigned long long)rl.rlim_cur, (unsigned long long)need);
    }
  }
  std::vector<Stream> st(S);
  const double td0 = now_s();
  for (int i = 0; i < S; ++i) st[i].fd_in = dial(host, port_in);
  const double td1 = now_s();
  usleep((useconds_t)(300000 + 100 * S));            // let the server adopt every connection before the next step
  meet("inputs");

  const double td2 = now_s();
  for (int i = 0; i < S; ++i) st[i].fd_out = dial(host, port_out);
  const double td3 = now_s();
  usleep((useconds_t)(300000 + 100 * S));
  meet("outputs");
  fprintf(stderr, "loadgen CLOCK_MONOTONIC: inputs dialled %.3f .. %.3f, outputs dialled %.3f .. %.3f\n", td0, td1, td2, td3);

  std::atomic<bool> stop{false};
  std::atomic<long> frames_sent{0}, frames_answered{0}, late{0}, slipped{0};
  std::atomic<long> max_send_lag_us{0}, max_recv_gap_us{0}, max_send_call_us{0};   // the load generator's own stalls (so they are not blamed on the server)
  auto amax = [](std::atomic<long>& a, long v) { long m = a.load(); while (v > m && !a.compare_exchange_weak(m, v)) {} };
  std::atomic<long> route_changes{0}, inband_bad{0};
  std::mutex lat_mu;
  struct Stall { double t; int stream; float ms; };
  std::vector<Stall> stalls;                          // answers later than 50 ms: when (s after the measurement start), which stream, how late
  std::vector<float> lats;
  lats.reserve((size_t)(S * hz * seconds * 1.1));
  double t_start = now_s() + 0.2;
  if (procs > 1) {                                   // rank 0 names the common start (CLOCK_MONOTONIC is system-wide), the others read it
    char p[512], q[512];
    snprintf(p, sizeof p, "%s/t_start", sync_dir);
    if (rank == 0) {
      snprintf(q, sizeof q, "%s/t_start.tmp", sync_dir);
      FILE* f = fopen(q, "w"); fprintf(f, "%.6f\n", now_s() + 1.0); fclose(f); rename(q, p);
    }
    FILE* f = nullptr;
    while (!(f = fopen(p, "r"))) usleep(2000);
    if (fscanf(f, "%lf", &t_start) != 1) { fprintf(stderr, "bad t_start\n"); return 2; }
    fclose(f);
  }
  fprintf(stderr, "loadgen CLOCK_MONOTONIC: t_start %.3f\n", t_start);
  const double t_measure = t_start + warm;          // latencies before this are not recorded (window fill / ramp-up)
  const double t_end = t_measure + seconds;

  auto sender = [&](int tid) {
    prctl(PR_SET_TIMERSLACK, 1000UL, 0UL, 0UL, 0UL);   // 1 us instead of the default 50 us: the sleeps above wake on time
    // streams tid, tid + threads, ...; stream i's frame k ends at t_start + (k + 1) * period + phase_i
    std::vector<int> mine;
    for (int i = tid; i < S; i += threads) mine.push_back(i);
    struct Ev { double t; int s; int pk; long frame; };
    auto later = [](const Ev& a, const Ev& b) { return a.t > b.t; };
    std::priority_queue<Ev, std::vector<Ev>, decltype(later)> q(later);
    for (int i : mine) q.push({t_start + period * ((long)i * procs + rank) / total_streams + period / packets_per_frame, i, 0, 0});
    std::vector<double> pkt((size_t)pk_samples * 2);
    while (!stop.load() && !q.empty()) {
      Ev e = q.top();
      q.pop();
      if (e.t >= t_end) continue;
      double n = now_s();
      if (n > e.t) amax(max_send_lag_us, (long)((n - e.t) * 1e6));
      if (n - e.t > 0.5 * period / packets_per_frame + 2e-3) { e.t = n; slipped.fetch_add(1); }   // fell behind: re-base — a microphone cannot burst either
      // Pace WITHOUT spinning: a packet may leave up to 150 us early (1.5 % of its 10 ms), anything further away is slept for (timer slack 1 us, set
      // at thread start).  The spin-wait this replaces burned a whole core per sender thread whatever the load - under a 16-core cgroup quota that
      // was cores taken from the server under test (round 6: 7 - 9 "load generator" cores at 8192 dialogues, most of it waiting).
      while (e.t - n > 150e-6) {
        const double until = e.t - 60e-6;
        timespec ts{(time_t)until, (long)((until - (double)(time_t)until) * 1e9)};
        clock_nanosleep(CLOCK_MONOTONIC, TIMER_ABSTIME, &ts, nullptr);
        n = now_s();
      }
      const uint8_t* p = (const uint8_t*)(audio.data() + ((size_t)(e.frame % NF) * hop + (size_t)e.pk * pk_samples) * 2);
      size_t left = (size_t)pk_samples * 16;
      if (e.pk + 1 == packets_per_frame) {   // time stamp BEFORE the frame's last packet leaves: the answer may overtake us
        const double ts = now_s();
        if (inband) {                          // the stamp and the dialogue id travel in the first sample pair of this packet
          memcpy(pkt.data(), p, left);
          pkt[0] = fmod(ts, 1024.0) * (1.0 / 1048576.0);
          pkt[1] = (double)((long)e.s * procs + rank + 1) * (1.0 / 1073741824.0);
          p = (const uint8_t*)pkt.data();
          if (st[e.s].frames++ == 0) st[e.s].t_first_sent = ts;
        } else {
          std::lock_guard<std::mutex> lk(st[e.s].mu);
          st[e.s].sent.push_back(ts);
          if (st[e.s].frames++ == 0) st[e.s].t_first_sent = st[e.s].sent.back();
        }
      }
      const double t_call = now_s();
      while (left) {
        ssize_t w = send(st[e.s].fd_in, p, left, MSG_NOSIGNAL);
        if (w <= 0) { if (errno == EINTR) continue; stop.store(true); break; }
        p += w; left -= (size_t)w;
      }
      amax(max_send_call_us, (long)((now_s() - t_call) * 1e6));
      if (++e.pk == packets_per_frame) {
        e.pk = 0;
        ++e.frame;
        frames_sent.fetch_add(1);
      }
      e.t += period / packets_per_frame;
      q.push(e);
    }
  };

  auto receiver = [&](int tid) {
    int ep = epoll_create1(0);
    for (int i = tid; i < S; i += threads) {
      epoll_event ev;
      memset(&ev, 0, sizeof ev);
      ev.events = EPOLLIN;
      ev.data.u32 = (uint32_t)i;
      epoll_ctl(ep, EPOLL_CTL_ADD, st[i].fd_out, &ev);
    }
    std::vector<uint8_t> buf(1 << 18);
    epoll_event evs[128];
    double t_busy = 0.0;
    while (!stop.load()) {
      int n = epoll_wait(ep, evs, 128, 50);
      if (n > 0 && t_busy > 0.0) amax(max_recv_gap_us, (long)((now_s() - t_busy) * 1e6));
      for (int k = 0; k < n; ++k) {
        Stream& s = st[evs[k].data.u32];
        ssize_t r = recv(s.fd_out, buf.data(), buf.size(), MSG_DONTWAIT);
        if (r <= 0) continue;
        const double t = now_s();
        s.rbuf.insert(s.rbuf.end(), buf.begin(), buf.begin() + r);
        size_t off = 0;
        while (s.rbuf.size() - off >= 4) {
          uint32_t len;
          memcpy(&len, s.rbuf.data() + off, 4);
          if (s.rbuf.size() - off < 4 + (size_t)len) break;
          double t_sent = 0;
          if (inband) {   // payload: f64 t | u32 n | x1[n] | u32 n | x2[n] | ...; the stamps sit at sample k0 = hop - pk_samples of x1 / x2
            const size_t k0 = (size_t)(hop - pk_samples), o1 = off + 4 + 8 + 4 + 8 * k0, o2 = off + 4 + 8 + 4 + 8 * (size_t)hop + 4 + 8 * k0;
            double v1 = 0, v2 = 0;
            if (len >= 8 + 4 + 8 * (size_t)hop + 4 + 8 * (size_t)hop) { memcpy(&v1, s.rbuf.data() + o1, 8); memcpy(&v2, s.rbuf.data() + o2, 8); }
            const long id = lround(v2 * 1073741824.0) - 1;
            if (id < 0 || v1 <= 0) inband_bad.fetch_add(1);
            else {
              const double tm = fmod(t, 1024.0), ts = v1 * 1048576.0;
              t_sent = t - fmod(tm - ts + 1024.0, 1024.0);
              if (s.heard >= 0 && s.heard != (int)id) route_changes.fetch_add(1);
              s.heard = (int)id;
            }
          } else {
            std::lock_guard<std::mutex> lk(s.mu);
            if (!s.sent.empty()) { t_sent = s.sent.front(); s.sent.pop_front(); }
          }
          off += 4 + len;
          if (s.answered == 0) s.t_first_answer = t;
          ++s.answered;
          frames_answered.fetch_add(1);
          if (t_sent >= t_measure) {
            const double ms = (t - t_sent) * 1e3;
            if (ms > late_ms) late.fetch_add(1);
            std::lock_guard<std::mutex> lk(lat_mu);
            lats.push_back((float)ms);
            if (ms > 50.0 && stalls.size() < 4096) stalls.push_back({t - t_measure, (int)evs[k].data.u32, (float)ms});
          }
        }
        if (off) s.rbuf.erase(s.rbuf.begin(), s.rbuf.begin() + off);
      }
      t_busy = n > 0 ? now_s() : 0.0;
    }
    close(ep);
  };

  std::vector<std::thread> th;
  for (int t = 0; t < threads; ++t) th.emplace_back(receiver, t);
  std::vector<std::thread> snd;
  for (int t = 0; t < threads; ++t) snd.emplace_back(sender, t);
  for (auto& t : snd) t.join();
  usleep(300000);                                    // let the last results arrive
  stop.store(true);
  for (auto& t : th) t.join();
  std::sort(lats.begin(), lats.end());
  auto pct = [&](double q) { return lats.empty() ? 0.0 : (double)lats[std::min(lats.size() - 1, (size_t)(q * lats.size()))]; };
  long unanswered = 0;
  for (auto& s : st) unanswered += (long)s.sent.size();
  if (inband) unanswered = frames_sent.load() - frames_answered.load();
  if (hist_out) {                                    // 50 us bins up to 400 ms (the last bin takes everything above)
    std::vector<long> hist(8001, 0);
    for (float v : lats) hist[std::min<size_t>(8000, (size_t)(v * 20.0f))]++;
    if (FILE* f = fopen(hist_out, "w")) {
      fprintf(f, "{\"bin_ms\": 0.05, \"samples\": %zu, \"max_ms\": %.3f, \"hist\": [", lats.size(), lats.empty() ? 0.0 : (double)lats.back());
      for (size_t i = 0; i < hist.size(); ++i) fprintf(f, "%s%ld", i ? "," : "", hist[i]);
      fprintf(f, "]}\n");
      fclose(f);
    }
  }
  // stall events: late answers clustered in time (a TCP retransmission timeout shows as one stream ~200 ms late, a host stall as many at once)
  std::sort(stalls.begin(), stalls.end(), [](const Stall& a, const Stall& b) { return a.t < b.t; });
  std::string stall_json = "[";
  {
    size_t i = 0; int shown = 0;
    while (i < stalls.size() && shown < 24) {
      size_t j = i; float worst = 0; std::vector<int> ids;
      while (j < stalls.size() && stalls[j].t - stalls[i].t < 0.3) { worst = std::max(worst, stalls[j].ms); if (std::find(ids.begin(), ids.end(), stalls[j].stream) == ids.end()) ids.push_back(stalls[j].stream); ++j; }
      char b[160];
      snprintf(b, sizeof b, "%s{\"t_s\": %.2f, \"late_answers\": %zu, \"streams\": %zu, \"first_stream\": %d, \"worst_ms\": %.1f}", shown ? ", " : "", stalls[i].t, j - i, ids.size(), ids[0], worst);
      stall_json += b; ++shown; i = j;
    }
    stall_json += "]";
  }
  printf("{\"stall_events_over_50ms\": %s, ", stall_json.c_str());
  {   // streams with unanswered frames: how many frames they sent / got back and when the first answer came (relative to their first frame)
    std::string u = "[";
    int shown = 0;
    for (int i = 0; i < S; ++i)
      if (!st[i].sent.empty() && shown < 24) {
        char b[160];
        snprintf(b, sizeof b, "%s{\"stream\": %d, \"frames_sent\": %ld, \"answered\": %ld, \"first_answer_after_first_frame_ms\": %.1f}", shown ? ", " : "", i, st[i].frames,
                 st[i].answered, (st[i].t_first_answer - st[i].t_first_sent) * 1e3);
        u += b; ++shown;
      }
    printf("\"streams_with_unanswered_frames\": %s], ", u.c_str());
  }
  printf("\"inband\": %d, \"procs\": %d, \"rank\": %d, \"route_changes\": %ld, \"inband_unreadable\": %ld, ", inband, procs, rank, route_changes.load(), inband_bad.load());
  printf("\"streams\": %d, \"frame_hz\": %d, \"packet_ms\": %d, \"seconds_measured\": %.1f, \"frames_sent\": %ld, \"frames_answered\": %ld, "
         "\"unanswered_at_end\": %ld, \"latency_samples\": %zu, \"lat_p50_ms\": %.3f, \"lat_p99_ms\": %.3f, \"lat_p999_ms\": %.3f, \"lat_max_ms\": %.3f, "
         "\"late_over_%.0fms\": %ld, \"schedule_slips\": %ld, \"client_max_send_lag_ms\": %.2f, \"client_max_send_call_ms\": %.2f, \"client_max_recv_pass_ms\": %.2f, \"stream_frames_per_s\": %.1f}\n",
         S, hz, packet_ms, seconds, frames_sent.load(), frames_answered.load(), unanswered, lats.size(), pct(0.50), pct(0.99), pct(0.999),
         lats.empty() ? 0.0 : (double)lats.back(), late_ms, late.load(), slipped.load(), max_send_lag_us.load() * 1e-3, max_send_call_us.load() * 1e-3,
         max_recv_gap_us.load() * 1e-3, lats.size() / seconds);
  for (auto& s : st) { close(s.fd_in); close(s.fd_out); }
  return 0;
}
