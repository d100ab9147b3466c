/* oracle/raster_ref.c -- scalar C restatement of gsplat==1.1.1 rasterization(packed=False), fwd + bwd.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): the shipped library never links or calls this.
 * It is the second, independent restatement of the rasterizer (the first is oracle/raster.py, plain
 * torch + autograd); tests require the two to agree, and bench.py times this one as the
 * `cpu_baseline` ("port": the build's CPU restatement -- the reference has no CPU path,
 * flow3d/scene_model.py:36,360).
 *
 * PARITY UNPINNED: gsplat (reference requirements.txt:137; call site flow3d/scene_model.py:360-373)
 * is an un-vendored CUDA-only dependency that cannot be installed here and the reference has no
 * tests; the algorithm below is gsplat 1.1.1's published one:
 *   fully_fused_projection_{fwd,bwd}, isect_tiles + sort + isect_offset_encode,
 *   rasterize_to_pixels_{fwd,bwd}.
 *
 * Build: make -C oracle   (two libraries: REAL=float -> libraster_ref_f32.so, double -> _f64.so)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifndef REAL
#define REAL float
#endif
#define TILE 16

typedef REAL real;

static void quat_to_rotmat(const real *q, real *R) {
  real n = (real)sqrt((double)(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]));
  real w = q[0] / n, x = q[1] / n, y = q[2] / n, z = q[3] / n;
  R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - w * z); R[2] = 2 * (x * z + w * y);
  R[3] = 2 * (x * y + w * z); R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - w * x);
  R[6] = 2 * (x * z - w * y); R[7] = 2 * (y * z + w * x); R[8] = 1 - 2 * (x * x + y * y);
}

static void mat3_mul(const real *A, const real *B, real *C) { /* C = A B */
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      real s = 0;
      for (int k = 0; k < 3; k++) s += A[i * 3 + k] * B[k * 3 + j];
      C[i * 3 + j] = s;
    }
}
static void mat3_mul_bt(const real *A, const real *B, real *C) { /* C = A B^T */
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      real s = 0;
      for (int k = 0; k < 3; k++) s += A[i * 3 + k] * B[j * 3 + k];
      C[i * 3 + j] = s;
    }
}
static void mat3_mul_at(const real *A, const real *B, real *C) { /* C = A^T B */
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      real s = 0;
      for (int k = 0; k < 3; k++) s += A[k * 3 + i] * B[k * 3 + j];
      C[i * 3 + j] = s;
    }
}

/* shared forward math of one Gaussian up to cov2d (before blur); returns 0 if near/far culled */
typedef struct {
  real R[9], M[9], cov[9], pc[3], covc[9], J[6], cov2d[4], tx, ty, rz;
  int in_x, in_y;
} proj_t;

static int proj_common(const real *mean, const real *quat, const real *scale, const real *V, const real *K,
                       int W, int H, real near_p, real far_p, proj_t *o) {
  real Rcw[9] = {V[0], V[1], V[2], V[4], V[5], V[6], V[8], V[9], V[10]};
  real t[3] = {V[3], V[7], V[11]};
  for (int i = 0; i < 3; i++)
    o->pc[i] = Rcw[i * 3] * mean[0] + Rcw[i * 3 + 1] * mean[1] + Rcw[i * 3 + 2] * mean[2] + t[i];
  if (o->pc[2] < near_p || o->pc[2] > far_p) return 0;
  quat_to_rotmat(quat, o->R);
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) o->M[i * 3 + j] = o->R[i * 3 + j] * scale[j];
  mat3_mul_bt(o->M, o->M, o->cov);
  real tmp[9];
  mat3_mul(Rcw, o->cov, tmp);
  mat3_mul_bt(tmp, Rcw, o->covc);
  real fx = K[0], fy = K[4];
  real x = o->pc[0], y = o->pc[1], z = o->pc[2];
  real tanx = (real)0.5 * W / fx, tany = (real)0.5 * H / fy;
  real limx = (real)1.3 * tanx, limy = (real)1.3 * tany;
  real rz = 1 / z, rz2 = rz * rz;
  real xr = x * rz, yr = y * rz;
  o->in_x = (xr <= limx && xr >= -limx);
  o->in_y = (yr <= limy && yr >= -limy);
  real cxr = xr < -limx ? -limx : (xr > limx ? limx : xr);
  real cyr = yr < -limy ? -limy : (yr > limy ? limy : yr);
  o->tx = z * cxr;
  o->ty = z * cyr;
  o->rz = rz;
  real *J = o->J;
  J[0] = fx * rz; J[1] = 0; J[2] = -fx * o->tx * rz2;
  J[3] = 0; J[4] = fy * rz; J[5] = -fy * o->ty * rz2;
  real JS[6];
  for (int i = 0; i < 2; i++)
    for (int j = 0; j < 3; j++) {
      real s = 0;
      for (int k = 0; k < 3; k++) s += J[i * 3 + k] * o->covc[k * 3 + j];
      JS[i * 3 + j] = s;
    }
  for (int i = 0; i < 2; i++)
    for (int j = 0; j < 2; j++) {
      real s = 0;
      for (int k = 0; k < 3; k++) s += JS[i * 3 + k] * J[j * 3 + k];
      o->cov2d[i * 2 + j] = s;
    }
  return 1;
}

/* fully_fused_projection_fwd.  V = viewmat row-major 4x4, K row-major 3x3. */
void ref_project_fwd(int N, const real *means, const real *quats, const real *scales, const real *V, const real *K,
                     int W, int H, real near_p, real far_p, real eps2d, real radius_clip, int32_t *radii,
                     real *means2d, real *depths, real *conics) {
  for (int g = 0; g < N; g++) {
    radii[g] = 0;
    means2d[2 * g] = means2d[2 * g + 1] = 0;
    depths[g] = 0;
    conics[3 * g] = conics[3 * g + 1] = conics[3 * g + 2] = 0;
    proj_t p;
    if (!proj_common(means + 3 * g, quats + 4 * g, scales + 3 * g, V, K, W, H, near_p, far_p, &p)) continue;
    real a = p.cov2d[0] + eps2d, b = p.cov2d[1], c = p.cov2d[3] + eps2d;
    real det = a * c - b * b;
    if (det <= 0) continue;
    real mid = (real)0.5 * (a + c);
    real disc = mid * mid - det;
    if (disc < (real)0.01) disc = (real)0.01;
    real v1 = mid + (real)sqrt((double)disc);
    real radius = (real)ceil((double)(3 * (real)sqrt((double)v1)));
    if (radius <= radius_clip) continue;
    real mx = K[0] * p.pc[0] * p.rz + K[2], my = K[4] * p.pc[1] * p.rz + K[5];
    if (mx + radius <= 0 || mx - radius >= W || my + radius <= 0 || my - radius >= H) continue;
    radii[g] = (int32_t)radius;
    means2d[2 * g] = mx;
    means2d[2 * g + 1] = my;
    depths[g] = p.pc[2];
    conics[3 * g] = c / det;
    conics[3 * g + 1] = -b / det;
    conics[3 * g + 2] = a / det;
  }
}

/* fully_fused_projection_bwd.  Gradients are ACCUMULATED into v_means/v_quats/v_scales/v_V(16). */
void ref_project_bwd(int N, const real *means, const real *quats, const real *scales, const real *V, const real *K,
                     int W, int H, real eps2d, const int32_t *radii, const real *conics, const real *v_means2d,
                     const real *v_depths, const real *v_conics, real *v_means, real *v_quats, real *v_scales,
                     real *v_V) {
  real Rcw[9] = {V[0], V[1], V[2], V[4], V[5], V[6], V[8], V[9], V[10]};
  real fx = K[0], fy = K[4];
  for (int g = 0; g < N; g++) {
    if (radii[g] <= 0) continue;
    proj_t p;
    proj_common(means + 3 * g, quats + 4 * g, scales + 3 * g, V, K, W, H, (real)-1e30, (real)1e30, &p);
    const real *mean = means + 3 * g, *q = quats + 4 * g, *sc = scales + 3 * g;
    real A = conics[3 * g], B = conics[3 * g + 1], C = conics[3 * g + 2];
    real vA = v_conics[3 * g], vB = (real)0.5 * v_conics[3 * g + 1], vC = v_conics[3 * g + 2];
    /* v_cov2d = -Minv v_Minv Minv */
    real t00 = A * vA + B * vB, t01 = A * vB + B * vC, t10 = B * vA + C * vB, t11 = B * vB + C * vC;
    real vc[4] = {-(t00 * A + t01 * B), -(t00 * B + t01 * C), -(t10 * A + t11 * B), -(t10 * B + t11 * C)};
    (void)eps2d; /* blur adds a constant: no gradient */
    real *J = p.J;
    /* v_covc = J^T vc J */
    real vcJ[6];
    for (int i = 0; i < 2; i++)
      for (int j = 0; j < 3; j++) vcJ[i * 3 + j] = vc[i * 2] * J[j] + vc[i * 2 + 1] * J[3 + j];
    real v_covc[9];
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) v_covc[i * 3 + j] = J[i] * vcJ[j] + J[3 + i] * vcJ[3 + j];
    /* v_J = vc J covc^T + vc^T J covc */
    real v_J[6];
    for (int i = 0; i < 2; i++)
      for (int j = 0; j < 3; j++) {
        real s = 0;
        for (int k = 0; k < 3; k++)
          s += vcJ[i * 3 + k] * p.covc[j * 3 + k] + (vc[i] * J[k] + vc[2 + i] * J[3 + k]) * p.covc[k * 3 + j];
        v_J[i * 3 + j] = s;
      }
    real x = p.pc[0], y = p.pc[1], rz = p.rz, rz2 = rz * rz, rz3 = rz2 * rz;
    real v_pc[3];
    v_pc[0] = fx * rz * v_means2d[2 * g];
    v_pc[1] = fy * rz * v_means2d[2 * g + 1];
    v_pc[2] = -(fx * x * v_means2d[2 * g] + fy * y * v_means2d[2 * g + 1]) * rz2 + v_depths[g];
    if (p.in_x) v_pc[0] += -fx * rz2 * v_J[2]; else v_pc[2] += -fx * rz3 * v_J[2] * p.tx;
    if (p.in_y) v_pc[1] += -fy * rz2 * v_J[5]; else v_pc[2] += -fy * rz3 * v_J[5] * p.ty;
    v_pc[2] += -fx * rz2 * v_J[0] - fy * rz2 * v_J[4] + 2 * fx * p.tx * rz3 * v_J[2] + 2 * fy * p.ty * rz3 * v_J[5];
    /* world: pc = Rcw mean + t ; covc = Rcw cov Rcw^T */
    for (int i = 0; i < 3; i++) {
      v_means[3 * g + i] += Rcw[i] * v_pc[0] + Rcw[3 + i] * v_pc[1] + Rcw[6 + i] * v_pc[2];
      for (int j = 0; j < 3; j++) v_V[i * 4 + j] += v_pc[i] * mean[j];
      v_V[i * 4 + 3] += v_pc[i];
    }
    real t1[9], t2[9], v_cov[9];
    mat3_mul(v_covc, Rcw, t1);      /* v_covc Rcw */
    mat3_mul_bt(t1, p.cov, t2);     /* v_covc Rcw cov^T */
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) v_V[i * 4 + j] += t2[i * 3 + j];
    real v_covcT[9];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) v_covcT[i * 3 + j] = v_covc[j * 3 + i];
    mat3_mul(v_covcT, Rcw, t1);
    mat3_mul(t1, p.cov, t2);        /* v_covc^T Rcw cov */
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) v_V[i * 4 + j] += t2[i * 3 + j];
    mat3_mul_at(Rcw, v_covc, t1);
    mat3_mul(t1, Rcw, v_cov);       /* Rcw^T v_covc Rcw */
    /* cov = M M^T : v_M = (v_cov + v_cov^T) M */
    real sym[9], v_M[9];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) sym[i * 3 + j] = v_cov[i * 3 + j] + v_cov[j * 3 + i];
    mat3_mul(sym, p.M, v_M);
    real vR[9];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) vR[i * 3 + j] = v_M[i * 3 + j] * sc[j];
    for (int j = 0; j < 3; j++)
      v_scales[3 * g + j] += p.R[j] * v_M[j] + p.R[3 + j] * v_M[3 + j] + p.R[6 + j] * v_M[6 + j];
    real n = (real)sqrt((double)(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]));
    real w = q[0] / n, qx = q[1] / n, qy = q[2] / n, qz = q[3] / n;
    real vq[4];
    vq[0] = 2 * (qx * (vR[7] - vR[5]) + qy * (vR[2] - vR[6]) + qz * (vR[3] - vR[1]));
    vq[1] = 2 * (-2 * qx * (vR[4] + vR[8]) + qy * (vR[3] + vR[1]) + qz * (vR[6] + vR[2]) + w * (vR[7] - vR[5]));
    vq[2] = 2 * (qx * (vR[3] + vR[1]) - 2 * qy * (vR[0] + vR[8]) + qz * (vR[7] + vR[5]) + w * (vR[2] - vR[6]));
    vq[3] = 2 * (qx * (vR[6] + vR[2]) + qy * (vR[7] + vR[5]) - 2 * qz * (vR[0] + vR[4]) + w * (vR[3] - vR[1]));
    real dot = vq[0] * w + vq[1] * qx + vq[2] * qy + vq[3] * qz;
    real qh[4] = {w, qx, qy, qz};
    for (int i = 0; i < 4; i++) v_quats[4 * g + i] += (vq[i] - dot * qh[i]) / n;
  }
}

/* ---------------------------------------------------------------------------------------------- */
static void tile_rect(real mx, real my, int32_t radius, int tw, int th, int *x0, int *y0, int *x1, int *y1) {
  float tx = (float)mx / TILE, ty = (float)my / TILE, tr = (float)radius / TILE;
  float fx0 = floorf(tx - tr), fy0 = floorf(ty - tr), fx1 = ceilf(tx + tr), fy1 = ceilf(ty + tr);
  *x0 = fx0 < 0 ? 0 : (fx0 > tw ? tw : (int)fx0);
  *y0 = fy0 < 0 ? 0 : (fy0 > th ? th : (int)fy0);
  *x1 = fx1 < 0 ? 0 : (fx1 > tw ? tw : (int)fx1);
  *y1 = fy1 < 0 ? 0 : (fy1 > th ? th : (int)fy1);
}

int64_t ref_isect_count(int N, const real *means2d, const int32_t *radii, int W, int H, int32_t *tiles_per_gauss) {
  int tw = (W + TILE - 1) / TILE, th = (H + TILE - 1) / TILE;
  int64_t tot = 0;
  for (int g = 0; g < N; g++) {
    int c = 0;
    if (radii[g] > 0) {
      int x0, y0, x1, y1;
      tile_rect(means2d[2 * g], means2d[2 * g + 1], radii[g], tw, th, &x0, &y0, &x1, &y1);
      c = (x1 - x0) * (y1 - y0);
    }
    tiles_per_gauss[g] = c;
    tot += c;
  }
  return tot;
}

typedef struct { int64_t key; int32_t gid; int32_t seq; } isect_t;
static int isect_cmp(const void *a, const void *b) {
  const isect_t *x = (const isect_t *)a, *y = (const isect_t *)b;
  if (x->key != y->key) return x->key < y->key ? -1 : 1;
  return x->seq < y->seq ? -1 : (x->seq > y->seq ? 1 : 0); /* stable: emission order */
}

/* isect_tiles + stable sort + isect_offset_encode; tile_offsets has tw*th+1 entries */
void ref_isect_sort(int N, const real *means2d, const int32_t *radii, const real *depths, int W, int H,
                    int64_t n_isect, int32_t *flatten_ids, int32_t *tile_offsets) {
  int tw = (W + TILE - 1) / TILE, th = (H + TILE - 1) / TILE;
  isect_t *buf = (isect_t *)malloc(sizeof(isect_t) * (size_t)(n_isect > 0 ? n_isect : 1));
  int64_t n = 0;
  for (int g = 0; g < N; g++) {
    if (radii[g] <= 0) continue;
    int x0, y0, x1, y1;
    tile_rect(means2d[2 * g], means2d[2 * g + 1], radii[g], tw, th, &x0, &y0, &x1, &y1);
    float d = (float)depths[g];
    int32_t bits;
    memcpy(&bits, &d, 4);
    for (int ty = y0; ty < y1; ty++)
      for (int tx = x0; tx < x1; tx++) {
        buf[n].key = ((int64_t)(ty * tw + tx) << 32) | (int64_t)(uint32_t)bits;
        buf[n].gid = g;
        buf[n].seq = (int32_t)n;
        n++;
      }
  }
  qsort(buf, (size_t)n, sizeof(isect_t), isect_cmp);
  int64_t pos = 0;
  for (int t = 0; t <= tw * th; t++) {
    while (pos < n && (buf[pos].key >> 32) < t) pos++;
    tile_offsets[t] = (int32_t)pos;
  }
  for (int64_t i = 0; i < n; i++) flatten_ids[i] = buf[i].gid;
  free(buf);
}

#define ALPHA_MIN ((real)(1.0 / 255.0))
#define ALPHA_MAX ((real)0.999)
#define T_STOP ((real)1e-4)

/* rasterize_to_pixels_fwd (per-pixel, no batching needed on a CPU).  background may be NULL. */
void ref_raster_fwd(int D, const real *means2d, const real *conics, const real *colors, const real *opac,
                    const real *background, int W, int H, const int32_t *flatten_ids, const int32_t *tile_offsets,
                    real *out, real *alphas, int32_t *last_ids) {
  int tw = (W + TILE - 1) / TILE;
  for (int i = 0; i < H; i++)
    for (int j = 0; j < W; j++) {
      int tile = (i / TILE) * tw + j / TILE;
      int s = tile_offsets[tile], e = tile_offsets[tile + 1];
      real px = j + (real)0.5, py = i + (real)0.5, T = 1;
      real *o = out + ((size_t)i * W + j) * D;
      for (int k = 0; k < D; k++) o[k] = 0;
      int cur = 0;
      for (int idx = s; idx < e; idx++) {
        int g = flatten_ids[idx];
        real dx = means2d[2 * g] - px, dy = means2d[2 * g + 1] - py;
        real sigma = (real)0.5 * (conics[3 * g] * dx * dx + conics[3 * g + 2] * dy * dy) + conics[3 * g + 1] * dx * dy;
        real alpha = opac[g] * (real)exp((double)-sigma);
        if (alpha > ALPHA_MAX) alpha = ALPHA_MAX;
        if (sigma < 0 || alpha < ALPHA_MIN) continue;
        real nT = T * (1 - alpha);
        if (nT <= T_STOP) break;
        real vis = alpha * T;
        for (int k = 0; k < D; k++) o[k] += colors[(size_t)g * D + k] * vis;
        cur = idx;
        T = nT;
      }
      alphas[(size_t)i * W + j] = 1 - T;
      if (background) for (int k = 0; k < D; k++) o[k] += T * background[k];
      last_ids[(size_t)i * W + j] = cur;
    }
}

/* rasterize_to_pixels_bwd; gradients ACCUMULATED into v_means2d [N,2], v_conics [N,3], v_colors [N,D], v_opac [N] */
void ref_raster_bwd(int D, const real *means2d, const real *conics, const real *colors, const real *opac,
                    const real *background, int W, int H, const int32_t *flatten_ids, const int32_t *tile_offsets,
                    const real *alphas, const int32_t *last_ids, const real *v_out, const real *v_alphas,
                    real *v_means2d, real *v_conics, real *v_colors, real *v_opac) {
  int tw = (W + TILE - 1) / TILE;
  real *buffer = (real *)malloc(sizeof(real) * (size_t)D);
  for (int i = 0; i < H; i++)
    for (int j = 0; j < W; j++) {
      int tile = (i / TILE) * tw + j / TILE;
      int s = tile_offsets[tile], e = tile_offsets[tile + 1];
      if (e <= s) continue;
      size_t pix = (size_t)i * W + j;
      real px = j + (real)0.5, py = i + (real)0.5;
      real T_final = 1 - alphas[pix], T = T_final;
      const real *vo = v_out + pix * D;
      real va = v_alphas[pix];
      for (int k = 0; k < D; k++) buffer[k] = 0;
      real bg_dot = 0;
      if (background) for (int k = 0; k < D; k++) bg_dot += background[k] * vo[k];
      for (int idx = last_ids[pix]; idx >= s; idx--) {
        int g = flatten_ids[idx];
        real dx = means2d[2 * g] - px, dy = means2d[2 * g + 1] - py;
        real A = conics[3 * g], B = conics[3 * g + 1], C = conics[3 * g + 2];
        real sigma = (real)0.5 * (A * dx * dx + C * dy * dy) + B * dx * dy;
        real vis = (real)exp((double)-sigma);
        real alpha = opac[g] * vis;
        if (alpha > ALPHA_MAX) alpha = ALPHA_MAX;
        if (sigma < 0 || alpha < ALPHA_MIN) continue;
        real ra = 1 / (1 - alpha);
        T *= ra;
        real fac = alpha * T;
        real v_alpha = 0;
        for (int k = 0; k < D; k++) {
          v_colors[(size_t)g * D + k] += fac * vo[k];
          v_alpha += (colors[(size_t)g * D + k] * T - buffer[k] * ra) * vo[k];
        }
        v_alpha += T_final * ra * va;
        if (background) v_alpha += -T_final * ra * bg_dot;
        if (opac[g] * vis <= ALPHA_MAX) {
          real v_sigma = -opac[g] * vis * v_alpha;
          v_conics[3 * g] += (real)0.5 * v_sigma * dx * dx;
          v_conics[3 * g + 1] += v_sigma * dx * dy;
          v_conics[3 * g + 2] += (real)0.5 * v_sigma * dy * dy;
          v_means2d[2 * g] += v_sigma * (A * dx + B * dy);
          v_means2d[2 * g + 1] += v_sigma * (B * dx + C * dy);
          v_opac[g] += vis * v_alpha;
        }
        for (int k = 0; k < D; k++) buffer[k] += colors[(size_t)g * D + k] * fac;
      }
    }
  free(buffer);
}
