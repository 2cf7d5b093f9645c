// Audio front end on the GPU: channel mean + polyphase sinc resampler, framed STFT (LDS FFT),
// sparse mel filterbank, log.  Replaces the torchaudio calls of AMTAPC_Extractor._wav2feature
// (etude/data/extractor.py:178-197), which the reference runs on the host CPU.
// All tables (resampling kernel, window, twiddles, mel filterbank in CSR form) are built by the
// host layer exactly as torchaudio builds them and handed over through etd_frontend_create.
#include "frontend.h"
#include "prof.h"

// ------------------------------------------------------------------------------------------------
// Resampler: y[j*new + p] = sum_k kern[p][k] * mono[j*orig + k - width]        (zero outside [0,L))
// Workgroup = 256 threads = phases; RB input blocks per workgroup share one LDS span.
// ------------------------------------------------------------------------------------------------
#define RB 8
__global__ __launch_bounds__(256) void k_resample(const float* __restrict__ wav, int channels, long long L,
                                                  const float* __restrict__ kernT /*[K][new]*/, int K, int width,
                                                  int orig, int nw, float* __restrict__ out, long long n_out) {
  extern __shared__ __attribute__((aligned(16))) float xs[];
  const long long j0 = (long long)blockIdx.x * RB;
  const int span = (RB - 1) * orig + K;
  const float invc = 1.f / (float)channels;
  for (int i = threadIdx.x; i < span; i += blockDim.x) {
    const long long src = j0 * orig + i - width;
    float v = 0.f;
    if (src >= 0 && src < L) {
      for (int c = 0; c < channels; ++c) v += wav[(long long)c * L + src];
      v *= invc;
    }
    xs[i] = v;
  }
  __syncthreads();
  for (int p = threadIdx.x; p < nw; p += blockDim.x) {
    float acc[RB];
#pragma unroll
    for (int j = 0; j < RB; ++j) acc[j] = 0.f;
    for (int k = 0; k < K; ++k) {
      const float w = kernT[(long long)k * nw + p];
#pragma unroll
      for (int j = 0; j < RB; ++j) acc[j] = fmaf(w, xs[j * orig + k], acc[j]);
    }
#pragma unroll
    for (int j = 0; j < RB; ++j) {
      const long long n = (j0 + j) * nw + p;
      if (n < n_out) out[n] = acc[j];
    }
  }
}

__global__ void k_mono(const float* __restrict__ wav, int channels, long long L, float* __restrict__ out) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long stride = (long long)gridDim.x * blockDim.x;
  const float invc = 1.f / (float)channels;
  for (; i < L; i += stride) {
    float v = 0.f;
    for (int c = 0; c < channels; ++c) v += wav[(long long)c * L + i];
    out[i] = v * invc;
  }
}

// ------------------------------------------------------------------------------------------------
// STFT frame -> power spectrum -> mel -> log.  One workgroup (256 threads) per frame, radix-2
// decimation-in-time FFT of the windowed real frame in LDS (n_fft = 2^lg, <= 4096).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_stft_mel(const float* __restrict__ x, long long N, int n_fft, int lg, int hop,
                                                  const float* __restrict__ window, const float2* __restrict__ tw /*[n_fft/2]*/,
                                                  const int* __restrict__ mel_start, const int* __restrict__ mel_len,
                                                  const int* __restrict__ mel_off, const float* __restrict__ mel_w, int n_mels,
                                                  float log_offset, float* __restrict__ feat, long long T, int pad_zero) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* re = sm;
  float* im = sm + n_fft;
  float* pw = sm + 2 * n_fft;                 // power spectrum [n_fft/2 + 1]
  const long long t = blockIdx.x;
  const int half = n_fft >> 1;
  // load + window + bit-reverse scatter (center=True; pad_mode "reflect", or "constant" = zeros: hft_transformer.py:130)
  for (int i = threadIdx.x; i < n_fft; i += blockDim.x) {
    long long idx = t * hop + i - half;
    bool inside = idx >= 0 && idx < N;
    if (!pad_zero) {
      if (idx < 0) idx = -idx;
      if (idx >= N) idx = 2 * (N - 1) - idx;
      inside = true;
    }
    const float v = inside ? x[idx] * window[i] : 0.f;
    const int rv = (int)(__brev((unsigned)i) >> (32 - lg));
    re[rv] = v;
    im[rv] = 0.f;
  }
  __syncthreads();
  for (int s = 0; s < lg; ++s) {
    const int hl = 1 << s;                     // half butterfly span
    for (int b = threadIdx.x; b < half; b += blockDim.x) {
      const int k = b & (hl - 1);
      const int i0 = ((b >> s) << (s + 1)) + k, i1 = i0 + hl;
      const float2 w = tw[k << (lg - 1 - s)];  // exp(-2*pi*i*k/(2*hl))
      const float xr = re[i1], xi = im[i1];
      const float tr = w.x * xr - w.y * xi, ti = w.x * xi + w.y * xr;
      const float ur = re[i0], ui = im[i0];
      re[i0] = ur + tr; im[i0] = ui + ti;
      re[i1] = ur - tr; im[i1] = ui - ti;
    }
    __syncthreads();
  }
  for (int f = threadIdx.x; f <= half; f += blockDim.x) pw[f] = re[f] * re[f] + im[f] * im[f];
  __syncthreads();
  for (int m = threadIdx.x; m < n_mels; m += blockDim.x) {
    const int s0 = mel_start[m], n = mel_len[m], o = mel_off[m];
    float acc = 0.f;
    for (int i = 0; i < n; ++i) acc = fmaf(pw[s0 + i], mel_w[o + i], acc);
    feat[t * n_mels + m] = logf(acc + log_offset);
  }
}

// ------------------------------------------------------------------------------------------------
// Frame-wise RMS energy (librosa.feature.rms, center=True, zero padding): frame t covers samples
// [t*hop - frame/2, t*hop + frame/2); one wave per frame.            etude/utils/preprocess.py:116-152
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_rms_frames(const float* __restrict__ x, long long N, int frame, int hop, float* __restrict__ out, long long T) {
  const int lane = threadIdx.x & 63;
  const long long t = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t >= T) return;
  const long long s0 = t * hop - frame / 2;
  float acc = 0.f;
  for (int i = lane; i < frame; i += 64) {
    const long long idx = s0 + i;
    const float v = (idx >= 0 && idx < N) ? x[idx] : 0.f;
    acc = fmaf(v, v, acc);
  }
  acc = wave_sum(acc);
  if (lane == 0) out[t] = sqrtf(acc / (float)frame);
}

extern "C" int etd_rms_frames(const float* x_dev, long long n, int frame_length, int hop_length, float* out_dev, long long n_frames, void* stream) {
  if (!x_dev || !out_dev || n < 0 || frame_length < 1 || hop_length < 1 || n_frames < 0 || n_frames > 1 + n / hop_length)
    ETD_FAIL(ETD_EINVAL, "rms_frames: bad arguments");
  if (n_frames == 0) return ETD_OK;
  hipLaunchKernelGGL(k_rms_frames, dim3((unsigned)((n_frames + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x_dev, n, frame_length, hop_length, out_dev, n_frames);
  HIP_TRY(hipGetLastError());
  return ETD_OK;
}

// ------------------------------------------------------------------------------------------------
struct etd_frontend {
  int sr_in, sr_out, orig, nw, K, width;
  int n_fft, lg, hop, n_mels;
  float log_offset;
  int pad_zero = 0;          // STFT centre padding: 0 reflect (torch.stft default), 1 zeros (pad_mode="constant")
  float* kernT = nullptr;
  float* window = nullptr;
  float2* tw = nullptr;
  int *mel_start = nullptr, *mel_len = nullptr, *mel_off = nullptr;
  float* mel_w = nullptr;
};

template <typename T>
static int upload(T** dst, const T* src, size_t n) {
  HIP_TRY(hipMalloc((void**)dst, n * sizeof(T) + 16));
  HIP_TRY(hipMemcpy(*dst, src, n * sizeof(T), hipMemcpyHostToDevice));
  return ETD_OK;
}

extern "C" int etd_frontend_create(int sr_in, int sr_out, int orig, int nw, int K, int width, const float* kernT_host,
                                   int n_fft, int hop, const float* window_host, int n_mels, const int* mel_start,
                                   const int* mel_len, const float* mel_w_host, float log_offset, etd_frontend** out) {
  if (!out || n_fft < 64 || n_fft > 4096 || (n_fft & (n_fft - 1)) || hop <= 0 || n_mels <= 0 || n_mels > 1024)
    ETD_FAIL(ETD_EINVAL, "frontend_create: bad n_fft/hop/n_mels");
  if (sr_in != sr_out && (!kernT_host || K <= 0 || nw <= 0 || orig <= 0)) ETD_FAIL(ETD_EINVAL, "frontend_create: bad resampler table");
  etd_frontend* f = new etd_frontend();
  f->sr_in = sr_in; f->sr_out = sr_out; f->orig = orig; f->nw = nw; f->K = K; f->width = width;
  f->n_fft = n_fft; f->hop = hop; f->n_mels = n_mels; f->log_offset = log_offset;
  f->lg = 0; while ((1 << f->lg) < n_fft) ++f->lg;
  if (sr_in != sr_out) ETD_TRY(upload(&f->kernT, kernT_host, (size_t)K * nw));
  ETD_TRY(upload(&f->window, window_host, (size_t)n_fft));
  {
    float2* tw = new float2[n_fft / 2];
    for (int k = 0; k < n_fft / 2; ++k) {
      const double ang = -2.0 * 3.14159265358979323846 * (double)k / (double)n_fft;
      tw[k] = make_float2((float)cos(ang), (float)sin(ang));
    }
    int rc = upload(&f->tw, tw, (size_t)n_fft / 2);
    delete[] tw;
    ETD_TRY(rc);
  }
  int* off = new int[n_mels];
  int tot = 0;
  for (int m = 0; m < n_mels; ++m) {
    if (mel_start[m] < 0 || mel_len[m] < 0 || mel_start[m] + mel_len[m] > n_fft / 2 + 1) { delete[] off; ETD_FAIL(ETD_EINVAL, "frontend_create: mel filter %d out of range", m); }
    off[m] = tot; tot += mel_len[m];
  }
  int rc = upload(&f->mel_start, mel_start, (size_t)n_mels);
  if (!rc) rc = upload(&f->mel_len, mel_len, (size_t)n_mels);
  if (!rc) rc = upload(&f->mel_off, off, (size_t)n_mels);
  if (!rc) rc = upload(&f->mel_w, mel_w_host, (size_t)(tot > 0 ? tot : 1));
  delete[] off;
  ETD_TRY(rc);
  *out = f;
  return ETD_OK;
}

extern "C" void etd_frontend_destroy(etd_frontend* f) {
  if (!f) return;
  (void)hipDeviceSynchronize();   // kernels of this handle may still be in flight
  (void)hipFree(f->kernT); (void)hipFree(f->window); (void)hipFree(f->tw);
  (void)hipFree(f->mel_start); (void)hipFree(f->mel_len); (void)hipFree(f->mel_off); (void)hipFree(f->mel_w);
  delete f;
}

extern "C" int etd_frontend_set_pad_mode(etd_frontend* f, int constant_zero) {
  if (!f || (constant_zero != 0 && constant_zero != 1)) ETD_FAIL(ETD_EINVAL, "frontend_set_pad_mode: bad args");
  f->pad_zero = constant_zero;
  return ETD_OK;
}

extern "C" long long etd_frontend_resampled_len(const etd_frontend* f, long long n_in) {
  if (f->sr_in == f->sr_out) return n_in;
  return ((long long)f->nw * n_in + f->orig - 1) / f->orig;      // ceil(new*L/orig)
}
extern "C" long long etd_frontend_num_frames(const etd_frontend* f, long long n_in) {
  return 1 + etd_frontend_resampled_len(f, n_in) / f->hop;
}

extern "C" int etd_frontend_run(etd_frontend* f, const float* wav_dev, int channels, long long n_in, float* resampled_dev,
                                float* feat_dev, long long feat_capacity_frames, long long* n_frames_out, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (!f || !wav_dev || !resampled_dev || channels <= 0 || n_in <= 0) ETD_FAIL(ETD_EINVAL, "frontend_run: bad args");
  const long long n16 = etd_frontend_resampled_len(f, n_in);
  const long long T = 1 + n16 / f->hop;
  if (feat_dev && !f->pad_zero && n16 <= f->n_fft / 2) ETD_FAIL(ETD_EINVAL, "frontend_run: clip shorter than n_fft/2 after resampling (reflect pad undefined)");
  if (feat_dev && T > feat_capacity_frames) ETD_FAIL(ETD_EINVAL, "frontend_run: feature buffer too small (%lld > %lld)", T, feat_capacity_frames);
  {
  ProfScope ps("k_resample", st, 2.0 * n16 * f->K, (double)n_in * channels * 4 + (double)n16 * 4);
  if (f->sr_in == f->sr_out) {
    hipLaunchKernelGGL(k_mono, dim3(2048), dim3(256), 0, st, wav_dev, channels, n_in, resampled_dev);
  } else {
    const long long nblk = (n16 + f->nw - 1) / f->nw;
    const int span = (RB - 1) * f->orig + f->K;
    hipLaunchKernelGGL(k_resample, dim3((unsigned)((nblk + RB - 1) / RB)), dim3(256), span * sizeof(float), st, wav_dev, channels, n_in,
                       f->kernT, f->K, f->width, f->orig, f->nw, resampled_dev, n16);
  }
  }
  if (!feat_dev) {                       // channel mean + resample only (analyze_volume needs no spectrogram)
    HIP_TRY(hipGetLastError());
    if (n_frames_out) *n_frames_out = 0;
    return ETD_OK;
  }
  ProfScope ps2("k_stft_mel", st, 0, (double)n16 * 4 + (double)T * f->n_mels * 4);
  const size_t sm = (size_t)(2 * f->n_fft + f->n_fft / 2 + 1) * sizeof(float);
  hipLaunchKernelGGL(k_stft_mel, dim3((unsigned)T), dim3(256), sm, st, resampled_dev, n16, f->n_fft, f->lg, f->hop, f->window, f->tw,
                     f->mel_start, f->mel_len, f->mel_off, f->mel_w, f->n_mels, f->log_offset, feat_dev, T, f->pad_zero);
  HIP_TRY(hipGetLastError());
  if (n_frames_out) *n_frames_out = T;
  return ETD_OK;
}
