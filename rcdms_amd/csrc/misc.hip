// misc.hip — the small kernels around the UNet body: timestep embedding, tiny-M linears, input
// assembly / layout conversion, fused CFG + DDIM update, weight repacking.
//
// Replaces (reference): diffusers Timesteps/TimestepEmbedding via src/models/unet.py:100-103,383-389;
// ResnetBlock3D.time_emb_proj src/models/resnet.py:191; the per-step torch.cat / chunk / scheduler.step
// sequence of src/pipelines/RCDMs_pipeline.py:482-497; einops "b c f h w -> (b f) c h w" resnet.py:14-16.
#include "common.h"

namespace {

__global__ void timestep_embed_kernel(const float* __restrict__ t, int rows, int dim, float* __restrict__ out) {
  const int half = dim / 2;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * half) return;
  const int r = idx / half, i = idx - r * half;
  // exponent = -ln(10000) * i / (half - freq_shift), freq_shift = 0; flip_sin_to_cos -> [cos | sin]
  const float w = expf(-9.210340371976184f * (float)i / (float)half);
  const float a = t[r] * w;
  out[(size_t)r * dim + i] = cosf(a);
  out[(size_t)r * dim + half + i] = sinf(a);
}

// one wave per output column n; rows <= 8 kept as static-indexed accumulators.  The (optionally SiLU'd) input rows
// are staged once per block in LDS (rows*K floats) and read back as float4, the weight row streams as 16-byte loads.
__global__ __launch_bounds__(256) void small_linear_kernel(const float* __restrict__ x, int rows, int K,
                                                           const f16* __restrict__ W, const float* __restrict__ bias,
                                                           int N, int silu_in, int silu_out, float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sx = (float*)smem;  // [rows][K]
  for (int i = threadIdx.x; i < rows * K; i += 256) {
    float v = x[i];
    if (silu_in) v = silu_f(v);
    sx[i] = v;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  for (int j = 0; j < 4; ++j) {  // 16 output columns per block: the staged input is reused 16 times
  const int n = blockIdx.x * 16 + (threadIdx.x >> 6) * 4 + j;
  if (n >= N) return;
  float acc[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) acc[r] = 0.f;
  for (int k = lane * 8; k < K; k += 512) {
    Pack16 w;
    w.u = *(const uint4*)(W + (size_t)n * K + k);
    float wf[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) wf[e] = (float)w.e[e];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      if (r < rows) {
        const f32x4 x0 = *(const f32x4*)(sx + r * K + k), x1 = *(const f32x4*)(sx + r * K + k + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          acc[r] = fmaf(x0[e], wf[e], acc[r]);
          acc[r] = fmaf(x1[e], wf[4 + e], acc[r]);
        }
      }
    }
  }
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    if (r < rows) {
      float v = wave_sum(acc[r]);
      if (lane == 0) {
        if (bias) v += bias[n];
        if (silu_out) v = silu_f(v);
        out[(size_t)r * N + n] = v;
      }
    }
  }
  }
}

// thread per output row (sample', frame, y, x): 9 gathered channels + zero pad.
__global__ void assemble_input_kernel(const float* __restrict__ lat, const float* __restrict__ mask,
                                      const float* __restrict__ masked, int S, int reps, int F, int H, int W, f16* out,
                                      int ld, int c_pad) {
  const size_t hw = (size_t)H * W, fhw = hw * F;
  const size_t total = (size_t)reps * S * fhw;
  const size_t m = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= total) return;
  const int sp = (int)(m / fhw);
  const size_t rem = m - (size_t)sp * fhw;  // f*hw + y*W + x
  const int s = sp % S;
  f16* o = out + m * ld;
#pragma unroll
  for (int c = 0; c < 4; ++c) o[c] = (f16)lat[((size_t)s * 4 + c) * fhw + rem];
  o[4] = (f16)mask[(size_t)sp * fhw + rem];
#pragma unroll
  for (int c = 0; c < 4; ++c) o[5 + c] = (f16)masked[((size_t)sp * 4 + c) * fhw + rem];
  for (int c = 9; c < c_pad; ++c) o[c] = (f16)0.f;
}

__global__ void ncfhw_to_rows_kernel(const float* __restrict__ x, int b, int C, int F, int H, int W, f16* out, int ld,
                                     int c_pad) {
  const size_t hw = (size_t)H * W, fhw = hw * F;
  const size_t total = (size_t)b * fhw;
  const size_t m = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= total) return;
  const int s = (int)(m / fhw);
  const size_t rem = m - (size_t)s * fhw;
  f16* o = out + m * ld;
  for (int c = 0; c < C; ++c) o[c] = (f16)x[((size_t)s * C + c) * fhw + rem];
  for (int c = C; c < c_pad; ++c) o[c] = (f16)0.f;
}

__global__ void rows_to_ncfhw_kernel(const f16* __restrict__ rows, int ld, int b, int C, int F, int H, int W,
                                     float* __restrict__ out) {
  const size_t hw = (size_t)H * W, fhw = hw * F;
  const size_t total = (size_t)b * C * fhw;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const size_t rem = idx % fhw;
  const size_t sc = idx / fhw;
  const int c = (int)(sc % C), s = (int)(sc / C);
  out[idx] = (float)rows[((size_t)s * fhw + rem) * ld + c];
}

// thread per latent element (s, c, f, y, x)
__global__ void cfg_ddim_kernel(const f16* __restrict__ eps, int ld, float* lat, int S, int reps, int F, int H, int W,
                                float gs, const float* __restrict__ coef, const int* __restrict__ step) {
  const size_t hw = (size_t)H * W, fhw = hw * F;
  const size_t total = (size_t)S * 4 * fhw;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const size_t rem = idx % fhw;
  const size_t sc = idx / fhw;
  const int c = (int)(sc & 3), s = (int)(sc >> 2);
  float e = (float)eps[((size_t)s * fhw + rem) * ld + c];
  if (reps == 2) {
    const float ec = (float)eps[((size_t)(S + s) * fhw + rem) * ld + c];
    e = e + gs * (ec - e);
  }
  const float* k = coef + (size_t)(*step) * 4;
  const float x = lat[idx];
  const float x0 = (x - k[1] * e) / k[0];
  lat[idx] = k[2] * x0 + k[3] * e;
}

__global__ void load_timestep_kernel(const float* __restrict__ ts, const int* __restrict__ step, float* t_out, int rows) {
  const int i = threadIdx.x;
  if (i < rows) t_out[i] = ts[*step];
}
__global__ void advance_step_kernel(int* step) {
  if (threadIdx.x == 0) *step = *step + 1;
}

__global__ void pack_f16_kernel(const float* __restrict__ src, f16* __restrict__ dst, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    dst[i] = (f16)src[i];
}
// dst[co][tap*cin_pad + ci] = w[co][ci][ky][kx], tap = ky*3+kx
__global__ void pack_conv3x3_kernel(const float* __restrict__ w, int c_out, int c_in, int cin_pad, f16* __restrict__ dst) {
  const size_t total = (size_t)c_out * 9 * cin_pad;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int ci = (int)(i % cin_pad);
    const size_t r = i / cin_pad;
    const int tap = (int)(r % 9), co = (int)(r / 9);
    dst[i] = ci < c_in ? (f16)w[((size_t)co * c_in + ci) * 9 + tap] : (f16)0.f;
  }
}
// packed row p: grp = p/64, j = p%64; source row = j<32 ? grp*32+j (hidden) : n_out/2 + grp*32 + (j-32) (gate)
__global__ void pack_geglu_kernel(const float* __restrict__ w, const float* __restrict__ bias, int n_out, int K,
                                  f16* __restrict__ wd, float* __restrict__ bd) {
  const size_t total = (size_t)n_out * K;
  const int half = n_out / 2;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int k = (int)(i % K), prow = (int)(i / K);
    const int grp = prow >> 6, j = prow & 63;
    const int src = j < 32 ? grp * 32 + j : half + grp * 32 + (j - 32);
    wd[i] = (f16)w[(size_t)src * K + k];
    if (k == 0 && bias && bd) bd[prow] = bias[src];
  }
}

inline int grid_for(size_t n, int block = 256, int cap = 8192) {
  size_t g = (n + block - 1) / block;
  if (g > (size_t)cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

}  // namespace

extern "C" {

int rcdm_timestep_embed(const float* t, int32_t rows, int32_t dim, float* out, void* stream) {
  if (!t || !out || rows <= 0 || dim <= 0 || (dim & 1)) return RCDM_EINVAL;
  const int n = rows * (dim / 2);
  hipLaunchKernelGGL(timestep_embed_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, t, rows, dim, out);
  return rcdm_check_launch();
}

int rcdm_small_linear(const float* x, int32_t rows, int32_t K, const void* W, const float* bias, int32_t N,
                      int32_t silu_in, int32_t silu_out, float* out, void* stream) {
  if (!x || !W || !out || rows <= 0 || K <= 0 || N <= 0) return RCDM_EINVAL;
  if (rows > 8 || (K & 7) || (size_t)rows * K * sizeof(float) > 64 * 1024) return RCDM_ESHAPE;
  hipLaunchKernelGGL(small_linear_kernel, dim3((N + 15) / 16), dim3(256), (size_t)rows * K * sizeof(float),
                     (hipStream_t)stream, x, rows, K,
                     (const f16*)W, bias, N, silu_in, silu_out, out);
  return rcdm_check_launch();
}

int rcdm_assemble_input(const float* latents, const float* mask, const float* masked, int32_t S, int32_t reps,
                        int32_t frames, int32_t H, int32_t W, void* out, int32_t ld, int32_t c_pad, void* stream) {
  if (!latents || !mask || !masked || !out) return RCDM_EINVAL;
  if (S <= 0 || (reps != 1 && reps != 2) || frames <= 0 || H <= 0 || W <= 0 || c_pad < 9 || ld < c_pad) return RCDM_EINVAL;
  const size_t total = (size_t)reps * S * frames * H * W;
  hipLaunchKernelGGL(assemble_input_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     latents, mask, masked, S, reps, frames, H, W, (f16*)out, ld, c_pad);
  return rcdm_check_launch();
}

int rcdm_ncfhw_to_rows(const float* x, int32_t b, int32_t C, int32_t frames, int32_t H, int32_t W, void* out, int32_t ld,
                       int32_t c_pad, void* stream) {
  if (!x || !out || b <= 0 || C <= 0 || frames <= 0 || H <= 0 || W <= 0 || c_pad < C || ld < c_pad) return RCDM_EINVAL;
  const size_t total = (size_t)b * frames * H * W;
  hipLaunchKernelGGL(ncfhw_to_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x,
                     b, C, frames, H, W, (f16*)out, ld, c_pad);
  return rcdm_check_launch();
}

int rcdm_rows_to_ncfhw(const void* rows, int32_t ld, int32_t b, int32_t C, int32_t frames, int32_t H, int32_t W,
                       float* out, void* stream) {
  if (!rows || !out || b <= 0 || C <= 0 || frames <= 0 || H <= 0 || W <= 0 || ld < C) return RCDM_EINVAL;
  const size_t total = (size_t)b * C * frames * H * W;
  hipLaunchKernelGGL(rows_to_ncfhw_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const f16*)rows, ld, b, C, frames, H, W, out);
  return rcdm_check_launch();
}

int rcdm_cfg_ddim_step(const void* eps, int32_t ld, float* latents, int32_t S, int32_t reps, int32_t frames, int32_t H,
                       int32_t W, float guidance_scale, const float* coef, const int32_t* step_counter, void* stream) {
  if (!eps || !latents || !coef || !step_counter) return RCDM_EINVAL;
  if (S <= 0 || (reps != 1 && reps != 2) || frames <= 0 || H <= 0 || W <= 0 || ld < 4) return RCDM_EINVAL;
  const size_t total = (size_t)S * 4 * frames * H * W;
  hipLaunchKernelGGL(cfg_ddim_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const f16*)eps, ld, latents, S, reps, frames, H, W, guidance_scale, coef, step_counter);
  return rcdm_check_launch();
}

int rcdm_load_timestep(const float* timesteps, const int32_t* step_counter, float* t_out, int32_t rows, void* stream) {
  if (!timesteps || !step_counter || !t_out || rows <= 0 || rows > 64) return RCDM_EINVAL;
  hipLaunchKernelGGL(load_timestep_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, timesteps, step_counter, t_out, rows);
  return rcdm_check_launch();
}

int rcdm_advance_step(int32_t* step_counter, void* stream) {
  if (!step_counter) return RCDM_EINVAL;
  hipLaunchKernelGGL(advance_step_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, step_counter);
  return rcdm_check_launch();
}

int rcdm_pack_f16(const float* src, void* dst, size_t n, void* stream) {
  if (!src || !dst) return RCDM_EINVAL;
  if (n == 0) return RCDM_OK;
  hipLaunchKernelGGL(pack_f16_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, src, (f16*)dst, n);
  return rcdm_check_launch();
}

int rcdm_pack_conv3x3(const float* w, int32_t c_out, int32_t c_in, int32_t cin_pad, void* dst, void* stream) {
  if (!w || !dst || c_out <= 0 || c_in <= 0 || cin_pad < c_in || (cin_pad & 7)) return RCDM_EINVAL;
  const size_t n = (size_t)c_out * 9 * cin_pad;
  hipLaunchKernelGGL(pack_conv3x3_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, w, c_out, c_in, cin_pad,
                     (f16*)dst);
  return rcdm_check_launch();
}

int rcdm_pack_geglu_rows(const float* w, const float* bias, int32_t n_out, int32_t K, void* w_dst, float* bias_dst,
                         void* stream) {
  if (!w || !w_dst || n_out <= 0 || K <= 0) return RCDM_EINVAL;
  if (n_out % 64) return RCDM_ESHAPE;
  const size_t n = (size_t)n_out * K;
  hipLaunchKernelGGL(pack_geglu_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, w, bias, n_out, K,
                     (f16*)w_dst, bias_dst);
  return rcdm_check_launch();
}

}  // extern "C"
