// Sanitizer harness for the native TCP front-end (vap-realtime_amd/csrc/ingest.cpp): the front-end is compiled INTO this
// program next to stubs of the few engine entry points it links against, opened over a step FUNCTION (vapx_ingest_open_fn),
// and driven by in-process clients through the phases that found its races in round 2: steady traffic, reconnect churn in
// the middle of frames, listeners that never read, and a close under load.  Built and run by tests/test_ingest_sanitizers.py
// with -fsanitize=thread and with -fsanitize=address,undefined; any report fails the test.
#include "../../vap-realtime_amd/csrc/ingest.cpp"

#include <atomic>
#include <cstdlib>

// ---- engine stubs (the function-backed front-end only needs the staging allocator) ----
extern "C" {
void* vapx_host_alloc(size_t bytes) { return calloc(1, bytes ? bytes : 1); }
void vapx_host_free(void* p) { free(p); }
int vapx_get_config(vapx_handle, vapx_config*) { return VAPX_E_INVAL; }
int vapx_reset_stream(vapx_handle, int32_t) { return VAPX_E_INVAL; }
int vapx_reset_carry(vapx_handle, int32_t) { return VAPX_E_INVAL; }
int vapx_step(vapx_handle, int32_t, const int32_t*, const float*, int32_t, float*, int32_t, void*) { return VAPX_E_INVAL; }
int32_t vapx_bad_slots(vapx_handle, int32_t*, int32_t) { return 0; }
const char* vapx_last_error(vapx_handle) { return "stub"; }
}

namespace {

constexpr int HZ = 20, HOP = 16000 / HZ, PACKET_SAMPLES = 160;
std::atomic<long> g_steps{0}, g_resets{0};

int step_fn(void*, int32_t n, const int32_t* ids, const float* audio, float* out) {
  for (int i = 0; i < n; ++i) {
    float* o = out + (size_t)i * VAPX_OUT_STRIDE;
    memset(o, 0, VAPX_OUT_STRIDE * sizeof(float));
    const float* a = audio + (size_t)i * 2 * HOP;
    o[0] = a[0]; o[1] = a[HOP];                       // p_now <- first sample of each channel: lets the client check routing
    o[2] = (float)ids[i]; o[3] = 0.5f;
    o[VAPX_OUT_NVALID] = 1.f;
  }
  g_steps.fetch_add(1);
  return 0;
}
void reset_fn(void*, int32_t) { g_resets.fetch_add(1); }

int dial(int port) {
  int s = socket(AF_INET, SOCK_STREAM, 0);
  sockaddr_in a;
  memset(&a, 0, sizeof a);
  a.sin_family = AF_INET;
  a.sin_port = htons((uint16_t)port);
  inet_pton(AF_INET, "127.0.0.1", &a.sin_addr);
  for (int t = 0; t < 200; ++t) {
    if (connect(s, (sockaddr*)&a, sizeof a) == 0) return s;
    usleep(5000);
  }
  perror("connect");
  exit(3);
}

bool send_all(int fd, const void* p, size_t n) {
  const char* c = (const char*)p;
  while (n) {
    ssize_t w = send(fd, c, n, MSG_NOSIGNAL);
    if (w <= 0) { if (errno == EINTR) continue; return false; }
    c += w; n -= (size_t)w;
  }
  return true;
}

// sends `frames` frames of a stream whose every sample of channel 1 is `tag` (10 ms packets like the reference client)
void send_frames(int fd, int frames, double tag, int stop_after_packets = -1) {
  std::vector<double> pk(PACKET_SAMPLES * 2);
  for (int i = 0; i < PACKET_SAMPLES; ++i) { pk[2 * i] = tag; pk[2 * i + 1] = -tag; }
  int sent = 0;
  for (int f = 0; f < frames; ++f)
    for (int p = 0; p < HOP / PACKET_SAMPLES; ++p) {
      if (stop_after_packets >= 0 && sent == stop_after_packets) return;
      if (!send_all(fd, pk.data(), pk.size() * 8)) return;
      ++sent;
    }
}

// reads result packets until `want` arrived or the peer closed / timed out; returns the count, checks the routing tag
int read_results(int fd, int want, double tag, std::atomic<long>* wrong) {
  timeval tv{5, 0};
  setsockopt(fd, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof tv);
  std::vector<uint8_t> buf;
  uint8_t tmp[65536];
  int got = 0;
  while (got < want) {
    ssize_t r = recv(fd, tmp, sizeof tmp, 0);
    if (r <= 0) break;
    buf.insert(buf.end(), tmp, tmp + r);
    size_t off = 0;
    while (buf.size() - off >= 4) {
      uint32_t len;
      memcpy(&len, buf.data() + off, 4);
      if (buf.size() - off < 4 + (size_t)len) break;
      // body: f64 t, u32 n, x1[n], u32 n, x2[n], u32 2, p_now[2], ...
      const uint8_t* b = buf.data() + off + 4;
      uint32_t n1;
      memcpy(&n1, b + 8, 4);
      double x0;
      memcpy(&x0, b + 12, 8);
      if (n1 != (uint32_t)HOP || x0 != tag) wrong->fetch_add(1);
      off += 4 + len;
      ++got;
    }
    if (off) buf.erase(buf.begin(), buf.begin() + off);
  }
  return got;
}

}  // namespace

int main() {
  const int S = 24, FRAMES = 40;
  vapx_ingest_config cfg;
  memset(&cfg, 0, sizeof cfg);
  cfg.struct_size = sizeof cfg;
  cfg.rx_threads = 3; cfg.tx_threads = 3; cfg.max_wait_us = 300; cfg.reset_on_connect = 1; cfg.gain = 1;
  vapx_ingest_handle g = nullptr;
  if (vapx_ingest_open_fn(step_fn, reset_fn, nullptr, S, S, HZ, VAPX_MODE_VAP, &cfg, &g) != 0) { fprintf(stderr, "open failed\n"); return 2; }
  int pin = 0, pout = 0;
  vapx_ingest_ports(g, &pin, &pout);
  std::atomic<long> wrong{0}, answered{0};

  // ---- phase 1: S dialogues, steady traffic, every frame answered and routed to its own listener ----
  {
    std::vector<int> fin(S), fout(S);
    for (int i = 0; i < S; ++i) fin[i] = dial(pin);
    usleep(100000);
    for (int i = 0; i < S; ++i) fout[i] = dial(pout);     // the k-th output connection listens to the k-th stream
    usleep(100000);
    std::vector<std::thread> th;
    for (int i = 0; i < S; ++i) {
      th.emplace_back([&, i] { send_frames(fin[i], FRAMES, 1.0 + i); });
      th.emplace_back([&, i] { answered.fetch_add(read_results(fout[i], FRAMES, 1.0 + i, &wrong)); });
    }
    for (auto& t : th) t.join();
    if (answered.load() != (long)S * FRAMES || wrong.load()) {
      fprintf(stderr, "phase 1: answered %ld of %d, %ld misrouted\n", answered.load(), S * FRAMES, wrong.load());
      return 4;
    }
    // ---- phase 2: churn — half of the senders die in the middle of a frame, come back, finish; listeners stay ----
    for (int round = 0; round < 3; ++round) {
      std::vector<std::thread> t2;
      answered.store(0);
      for (int i = 0; i < S; i += 2) {
        t2.emplace_back([&, i, round] {
          send_frames(fin[i], 3, 1.0 + i, 7 + round);        // 1 frame + part of the next, then vanish
          close(fin[i]);
          fin[i] = dial(pin);                                 // a reconnect takes the lowest free slot: with all others busy, slot i again
          send_frames(fin[i], 10, 1.0 + i);
        });
      }
      for (auto& t : t2) t.join();
      usleep(200000);
    }
    // ---- phase 3: listeners that never read + a burst: the front-end must drop them, not stall the others ----
    std::vector<int> lazy;
    for (int i = 0; i < 4; ++i) lazy.push_back(dial(pout));
    {
      std::vector<std::thread> t3;
      for (int i = 0; i < S; ++i) t3.emplace_back([&, i] { send_frames(fin[i], 400, 1.0 + i); });   // ~5 MB per listener: nobody reads any more
      for (auto& t : t3) t.join();
    }
    // ---- phase 4: close while a third of the senders are still pushing ----
    std::atomic<bool> stop{false};
    std::vector<std::thread> t4;
    for (int i = 0; i < S; i += 3) t4.emplace_back([&, i] { while (!stop.load()) send_frames(fin[i], 2, 1.0 + i); });
    usleep(150000);
    vapx_ingest_stats st;
    vapx_ingest_stats_read(g, &st, 1);
    vapx_ingest_close(g);
    stop.store(true);
    for (auto& t : t4) t.join();
    for (int fd : fin) close(fd);
    for (int fd : fout) close(fd);
    for (int fd : lazy) close(fd);
    printf("ok: phase 1 %d frames answered, steps %ld, resets %ld, frames_done %ld, dropped listeners %ld\n", S * FRAMES, g_steps.load(),
           g_resets.load(), (long)st.frames_done, (long)st.dropped_listeners);
  }
  return 0;
}
