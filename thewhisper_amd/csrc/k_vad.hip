// Voice-activity gate for the streaming scheduler.
//
// The reference gates `StreamingPipeline.add_new_chunk` with silero-vad fetched through torch.hub
// (R:thestage_speechkit/streaming/streaming_pipeline.py:533-538) and consumes it as
// `prob = vad_model(frame_512, 16000).item(); speech = prob > vad_threshold` on consecutive 512-sample frames, the model
// keeping state between calls (:589-622).  The silero weights are not obtainable offline, so this is NOT a port of that
// network: it is an adaptive-noise-floor energy detector with the same calling contract (stateful, one probability per
// 512-sample frame), stated here and restated in oracle/whisper_oracle.py::energy_vad:
//
//     e   = 10 log10( mean(x^2) + 1e-10 )                      frame level in dB (sum of squares in float64)
//     nf  = min(e, -40)                       on the first frame of a stream
//     nf  = min(e, nf + 0.02)                 afterwards: the floor follows drops at once and creeps up 0.02 dB / frame
//     p   = 1 / (1 + exp(-((e - nf) - 9) / 2))   if e > -60 dB, else 0
//
// One wavefront per stream walks that stream's frames in order (the state is sequential); streams are independent, so
// a serving tick runs the frames of all sessions in ONE launch.
#include "tw_common.h"

namespace {

__global__ __launch_bounds__(64) void vad_energy_kernel(const float* __restrict__ pcm, long long stream_stride, int n_frames,
                                                         float* __restrict__ state, float* __restrict__ prob) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const float* x = pcm + (long long)b * stream_stride;
  float nf = state[2 * b];
  bool started = state[2 * b + 1] != 0.f;
  for (int f = 0; f < n_frames; ++f) {
    const float* fr = x + (long long)f * 512;
    const f32x4_t a = *reinterpret_cast<const f32x4_t*>(fr + lane * 8);
    const f32x4_t c = *reinterpret_cast<const f32x4_t*>(fr + lane * 8 + 4);
    double s = (double)a[0] * a[0] + (double)a[1] * a[1] + (double)a[2] * a[2] + (double)a[3] * a[3] + (double)c[0] * c[0] +
               (double)c[1] * c[1] + (double)c[2] * c[2] + (double)c[3] * c[3];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    const float e = 10.0f * log10f((float)(s * (1.0 / 512.0)) + 1e-10f);
    nf = started ? fminf(e, nf + 0.02f) : fminf(e, -40.0f);
    started = true;
    const float p = (e > -60.0f) ? 1.0f / (1.0f + expf(-((e - nf) - 9.0f) * 0.5f)) : 0.0f;
    if (lane == 0) prob[(long long)b * n_frames + f] = p;
  }
  if (lane == 0) {
    state[2 * b] = nf;
    state[2 * b + 1] = started ? 1.f : 0.f;
  }
}

}  // namespace

hipError_t launch_vad_energy(const float* pcm, long long stream_stride, int B, int n_frames, float* state, float* prob,
                             hipStream_t st) {
  if (B < 1 || n_frames < 1) return hipErrorInvalidValue;
  hipLaunchKernelGGL(vad_energy_kernel, dim3(B), dim3(64), 0, st, pcm, stream_stride, n_frames, state, prob);
  return hipGetLastError();
}
