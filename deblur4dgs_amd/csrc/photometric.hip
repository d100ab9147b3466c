// photometric.hip -- SURVEY 8f-2: the photometric term of the training loss, fused.
//
// Reference: flow3d/trainer.py:388-392,575-586
//     0.8 * F.l1_loss(pred * m, gt * m) + 0.2 * (1 - SSIM(pred * m, gt * m)),   SSIM = pytorch_msssim.SSIM(1.0, channel=3)
// (pytorch-msssim 1.0.0: 11-tap sigma-1.5 Gaussian, separable, no padding; see oracle/photometric.py).  In eager
// PyTorch one evaluation is ~25 launches forward and ~60 backward, three to four times per step; here it is one
// tile kernel forward (loss partials + the three derivative maps dS/dmu1, dS/dE[x^2], dS/dE[xy]), a one-block ordered
// sum, and one tile kernel backward (transposed separable filter of the maps + the L1 sign term).  Images are read
// channel-last [B,H,W,C] exactly as the rasterizer writes them; the mask [B,H,W] multiplies both images.
#include "common.h"

namespace {

constexpr int PW = 11, PT = 16, PH = PT + PW - 1;  // window, tile, tile + halo (26)
constexpr int PC = 3;                                      // channels (the reference's SSIM is built for 3)
__constant__ float c_win[PW] = {0.00102838f, 0.00759876f, 0.03600077f, 0.10936069f, 0.21300554f, 0.26601172f,
                                0.21300554f, 0.10936069f, 0.03600077f, 0.00759876f, 0.00102838f};
constexpr float SSIM_C1 = 1e-4f, SSIM_C2 = 9e-4f;

struct PhotoArgs {
  const float *pred, *gt, *mask;  // [B,H,W,C], [B,H,W,C], [B,H,W] or null
  int B, H, W, Ho, Wo, tiles_x, tiles_y;
  float *maps;      // [B,Ho,Wo,C,3]
  float *partials;  // [n_blocks,2]  {sum of ssim_map, sum of |x - y|}
};

__global__ void __launch_bounds__(256) k_photo_fwd(const PhotoArgs a) {
  __shared__ float sx[PH * PH * PC], sy[PH * PH * PC];
  __shared__ float hbuf[5 * PH * PT * PC];  // horizontally filtered x, y, xx, yy, xy
  __shared__ float red[2 * 4];
  const int tid = threadIdx.x;
  const int b = blockIdx.z, ty0 = blockIdx.y * PT, tx0 = blockIdx.x * PT;
  // stage the 26x26 input patch (masked).  Tiles cover the INPUT grid, so every input pixel's |x - y| is owned by
  // exactly one block (its 16x16 top-left cells); output pixels exist only for oy < H - 10, ox < W - 10.
  float l1 = 0.f;
  const int own_h = PT, own_w = PT;
  for (int i = tid; i < PH * PH; i += 256) {
    const int ly = i / PH, lx = i - ly * PH;
    const int y = ty0 + ly, x = tx0 + lx;
    float m = 0.f;
    float px[PC] = {0.f, 0.f, 0.f}, gx[PC] = {0.f, 0.f, 0.f};
    if (y < a.H && x < a.W) {
      const size_t p = ((size_t)b * a.H + y) * a.W + x;
      m = a.mask ? a.mask[p] : 1.f;
#pragma unroll
      for (int c = 0; c < PC; c++) px[c] = a.pred[p * PC + c] * m, gx[c] = a.gt[p * PC + c] * m;
      if (ly < own_h && lx < own_w) {
#pragma unroll
        for (int c = 0; c < PC; c++) l1 += fabsf(px[c] - gx[c]);
      }
    }
#pragma unroll
    for (int c = 0; c < PC; c++) sx[i * PC + c] = px[c], sy[i * PC + c] = gx[c];
  }
  __syncthreads();
  // horizontal pass: 26 rows x 16 columns x 3 channels
  for (int i = tid; i < PH * PT * PC; i += 256) {
    const int c = i % PC, lx = (i / PC) % PT, ly = i / (PC * PT);
    float hx = 0.f, hy = 0.f, hxx = 0.f, hyy = 0.f, hxy = 0.f;
#pragma unroll
    for (int k = 0; k < PW; k++) {
      const float w = c_win[k], xv = sx[(ly * PH + lx + k) * PC + c], yv = sy[(ly * PH + lx + k) * PC + c];
      hx += w * xv, hy += w * yv, hxx += w * xv * xv, hyy += w * yv * yv, hxy += w * xv * yv;
    }
    hbuf[0 * PH * PT * PC + i] = hx, hbuf[1 * PH * PT * PC + i] = hy, hbuf[2 * PH * PT * PC + i] = hxx;
    hbuf[3 * PH * PT * PC + i] = hyy, hbuf[4 * PH * PT * PC + i] = hxy;
  }
  __syncthreads();
  // vertical pass + SSIM map + derivative maps: one thread per output pixel, looping channels
  const int ly = tid / PT, lx = tid % PT;
  const int oy = ty0 + ly, ox = tx0 + lx;
  float ssum = 0.f;
  if (oy < a.Ho && ox < a.Wo) {
#pragma unroll
    for (int c = 0; c < PC; c++) {
      float mu1 = 0.f, mu2 = 0.f, e11 = 0.f, e22 = 0.f, e12 = 0.f;
#pragma unroll
      for (int k = 0; k < PW; k++) {
        const float w = c_win[k];
        const int i = ((ly + k) * PT + lx) * PC + c;
        mu1 += w * hbuf[i], mu2 += w * hbuf[PH * PT * PC + i], e11 += w * hbuf[2 * PH * PT * PC + i];
        e22 += w * hbuf[3 * PH * PT * PC + i], e12 += w * hbuf[4 * PH * PT * PC + i];
      }
      const float s1 = e11 - mu1 * mu1, s2 = e22 - mu2 * mu2, s12 = e12 - mu1 * mu2;
      const float A1 = 2.f * mu1 * mu2 + SSIM_C1, A2 = 2.f * s12 + SSIM_C2;
      const float B1 = mu1 * mu1 + mu2 * mu2 + SSIM_C1, B2 = s1 + s2 + SSIM_C2;
      const float iB = 1.f / (B1 * B2);
      const float S = A1 * A2 * iB;
      ssum += S;
      // derivatives w.r.t. mu1, E[x^2], E[xy] (sigma's expanded: s1 = e11 - mu1^2, s12 = e12 - mu1 mu2)
      float *mp = a.maps + ((((size_t)b * a.Ho + oy) * a.Wo + ox) * PC + c) * 3;
      mp[0] = 2.f * mu2 * (A2 - A1) * iB - S * 2.f * mu1 * (B2 - B1) * iB;
      mp[1] = -S / B2;
      mp[2] = 2.f * A1 * iB;
    }
  }
  // block sums in a fixed order: lanes (shuffle tree), then the 4 waves
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) ssum += __shfl_xor(ssum, o), l1 += __shfl_xor(l1, o);
  if ((tid & 63) == 0) red[tid >> 6] = ssum, red[4 + (tid >> 6)] = l1;
  __syncthreads();
  if (tid == 0) {
    const int blk = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    a.partials[blk * 2] = (red[0] + red[1]) + (red[2] + red[3]);
    a.partials[blk * 2 + 1] = (red[4] + red[5]) + (red[6] + red[7]);
  }
}

// loss[0] = w_l1 * l1 + w_ssim * (1 - ssim), loss[1] = l1, loss[2] = ssim; one block, ordered
__global__ void __launch_bounds__(256) k_photo_finish(const float *partials, int n_blocks, float inv_nssim, float inv_nl1,
                                                      float w_l1, float w_ssim, float *loss) {
  __shared__ double rs[256], rl[256];
  double s = 0.0, l = 0.0;
  for (int i = threadIdx.x; i < n_blocks; i += 256) s += partials[2 * i], l += partials[2 * i + 1];
  rs[threadIdx.x] = s, rl[threadIdx.x] = l;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) rs[threadIdx.x] += rs[threadIdx.x + o], rl[threadIdx.x] += rl[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float ssim = (float)(rs[0] * inv_nssim), l1 = (float)(rl[0] * inv_nl1);
    loss[0] = w_l1 * l1 + w_ssim * (1.f - ssim), loss[1] = l1, loss[2] = ssim;
  }
}

struct PhotoBwdArgs {
  const float *pred, *gt, *mask, *maps, *v_loss;
  int B, H, W, Ho, Wo;
  float k_ssim, k_l1;  // -w_ssim / n_ssim, w_l1 / n_l1
  float *v_pred;
};

// dL/dpred(q) = m(q) * v * [ k_l1 sign(x - y) + k_ssim * sum_p w(p - q) (Da(p) + 2 x(q) D11(p) + y(q) D12(p)) ]
__global__ void __launch_bounds__(256) k_photo_bwd(const PhotoBwdArgs a) {
  __shared__ float sm[PH * PH * PC * 3];  // maps patch: output pixels q - 10 .. q (zero outside the valid region)
  __shared__ float hb[PH * PT * PC * 3];
  const int tid = threadIdx.x;
  const int b = blockIdx.z, ty0 = blockIdx.y * PT, tx0 = blockIdx.x * PT;
  for (int i = tid; i < PH * PH; i += 256) {
    const int ly = i / PH, lx = i - ly * PH;
    const int oy = ty0 + ly - (PW - 1), ox = tx0 + lx - (PW - 1);
    const bool in = oy >= 0 && ox >= 0 && oy < a.Ho && ox < a.Wo;
    const float *mp = a.maps + (((size_t)b * a.Ho + (in ? oy : 0)) * a.Wo + (in ? ox : 0)) * PC * 3;
#pragma unroll
    for (int j = 0; j < PC * 3; j++) sm[i * PC * 3 + j] = in ? mp[j] : 0.f;
  }
  __syncthreads();
  // horizontal (transposed): input column x gathers output columns x - 10 .. x with weight w[x - ox]
  for (int i = tid; i < PH * PT * PC * 3; i += 256) {
    const int j = i % (PC * 3), lx = (i / (PC * 3)) % PT, ly = i / (PC * 3 * PT);
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < PW; k++) acc += c_win[PW - 1 - k] * sm[(ly * PH + lx + k) * PC * 3 + j];
    hb[i] = acc;
  }
  __syncthreads();
  const int ly = tid / PT, lx = tid % PT;
  const int y = ty0 + ly, x = tx0 + lx;
  if (y >= a.H || x >= a.W) return;
  const size_t p = ((size_t)b * a.H + y) * a.W + x;
  const float m = a.mask ? a.mask[p] : 1.f;
  const float v = a.v_loss[0];
#pragma unroll
  for (int c = 0; c < PC; c++) {
    float da = 0.f, d11 = 0.f, d12 = 0.f;
#pragma unroll
    for (int k = 0; k < PW; k++) {
      const float w = c_win[PW - 1 - k];
      const float *h = hb + (((ly + k) * PT + lx) * PC + c) * 3;
      da += w * h[0], d11 += w * h[1], d12 += w * h[2];
    }
    const float xv = a.pred[p * PC + c] * m, yv = a.gt[p * PC + c] * m;
    const float sgn = xv > yv ? 1.f : (xv < yv ? -1.f : 0.f);
    a.v_pred[p * PC + c] = m * v * (a.k_l1 * sgn + a.k_ssim * (da + 2.f * xv * d11 + yv * d12));
  }
}

}  // namespace

int d4gs_photometric_fwd_impl(const float *pred, const float *gt, const float *mask, int32_t B, int32_t H, int32_t W,
                              float w_l1, float w_ssim, float *maps, float *partials, float *loss, hipStream_t stream) {
  PhotoArgs a;
  a.pred = pred, a.gt = gt, a.mask = mask, a.B = B, a.H = H, a.W = W, a.Ho = H - (PW - 1), a.Wo = W - (PW - 1);
  a.maps = maps, a.partials = partials;
  a.tiles_x = (W + PT - 1) / PT, a.tiles_y = (H + PT - 1) / PT;
  const dim3 grid(a.tiles_x, a.tiles_y, B);
  {
    ProfScope ps("k_photo_fwd", stream);
    k_photo_fwd<<<grid, 256, 0, stream>>>(a);
  }
  int rc = d4gs_check_launch("k_photo_fwd");
  if (rc) return rc;
  const int nb = a.tiles_x * a.tiles_y * B;
  ProfScope ps("k_photo_finish", stream);
  k_photo_finish<<<1, 256, 0, stream>>>(partials, nb, 1.f / ((float)B * PC * a.Ho * a.Wo), 1.f / ((float)B * PC * H * W), w_l1,
                                        w_ssim, loss);
  return d4gs_check_launch("k_photo_finish");
}

int d4gs_photometric_bwd_impl(const float *pred, const float *gt, const float *mask, const float *maps,
                              const float *v_loss, int32_t B, int32_t H, int32_t W, float w_l1, float w_ssim,
                              float *v_pred, hipStream_t stream) {
  PhotoBwdArgs a;
  a.pred = pred, a.gt = gt, a.mask = mask, a.maps = maps, a.v_loss = v_loss, a.B = B, a.H = H, a.W = W;
  a.Ho = H - (PW - 1), a.Wo = W - (PW - 1);
  a.k_ssim = -w_ssim / ((float)B * PC * a.Ho * a.Wo), a.k_l1 = w_l1 / ((float)B * PC * H * W);
  a.v_pred = v_pred;
  const dim3 grid((W + PT - 1) / PT, (H + PT - 1) / PT, B);
  ProfScope ps("k_photo_bwd", stream);
  k_photo_bwd<<<grid, 256, 0, stream>>>(a);
  return d4gs_check_launch("k_photo_bwd");
}

extern "C" int64_t d4gs_photometric_blocks(int32_t B, int32_t H, int32_t W) {
  return (int64_t)B * ((W + PT - 1) / PT) * ((H + PT - 1) / PT);
}
