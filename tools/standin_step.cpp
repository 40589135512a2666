// Stand-in for the GPU engine behind the native front-end (vapx_ingest_open_fn): the host-side half of BASELINE config 4 (8 x 4096 dialogues
// behind ONE front door) can then be load-tested on a box without eight GPUs.  A tick of the real engine costs the host nothing but the wait
// (the audio block is DMA'd from the pinned staging, the results come back the same way), so the stand-in (a) touches the audio and writes a
// result row per stream like vapx_step's host path does and (b) SLEEPS for the tick time measured with the real engine behind the same
// front-end: base_us + per_stream_ns * n (profiles/r05_frontend: 1.72 ms at a mean batch of 165 streams, 23 ms for a whole-batch tick of 4096
// at 20 Hz -> 800 us + 5.6 us per stream).
//   g++ -O2 -shared -fPIC -o tools/libstandin_step.so tools/standin_step.cpp      (make -C vap-realtime_amd/csrc tools)
#include <math.h>
#include <stdint.h>
#include <time.h>

extern "C" {
struct standin_cfg {
  int32_t hop;            // samples per channel per frame (16000 / frame_hz)
  int32_t out_stride;     // VAPX_OUT_STRIDE
  int32_t base_us;        // tick time = base_us + per_stream_ns * n / 1000
  int32_t per_stream_ns;
  int64_t ticks, frames;  // counters (written by the tick thread only)
};

int standin_step(void* user, int32_t n, const int32_t* ids, const float* audio, float* out) {
  standin_cfg* c = (standin_cfg*)user;
  for (int32_t k = 0; k < n; ++k) {
    const float* a = audio + (long)k * 2 * c->hop;
    float m0 = 0.f, m1 = 0.f;
    for (int i = 0; i < c->hop; i += 16) { m0 += fabsf(a[i]); m1 += fabsf(a[c->hop + i]); }   // one cache line in four of the frame, both channels
    float* o = out + (long)k * c->out_stride;
    const float s = m0 + m1 + 1e-9f;
    o[0] = m0 / s; o[1] = m1 / s;            // p_now
    o[2] = m1 / s; o[3] = m0 / s;            // p_future
    o[4] = m0 > 0.5f; o[5] = m1 > 0.5f;      // vad
    o[13] = 0.f;                             // VAPX_OUT_STATUS: finite
    (void)ids;
  }
  const long ns = (long)c->base_us * 1000 + (long)c->per_stream_ns * n;
  timespec ts{ns / 1000000000L, ns % 1000000000L};
  clock_nanosleep(CLOCK_MONOTONIC, 0, &ts, nullptr);
  ++c->ticks; c->frames += n;
  return 0;
}
}
