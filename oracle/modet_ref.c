/* TEST INFRASTRUCTURE ONLY -- plain-C (fp64) restatement of the ModeT hot-path arithmetic.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use anything under oracle/;
 * the product path (smilecode_amd/) never links or calls it.
 * Straight loop nests, no dependencies; every function names the reference lines it follows
 * (paths relative to /root/reference).  Parity pin: tests/test_cpu.py checks each function against the golden
 * vectors captured from the real reference (tests/golden/*.npz) -- the reference itself ships no tests.
 * Build: oracle/build_cref.py (gcc -O2 -shared -fPIC) -> oracle/_build/libmodet_ref.so.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define IDX3(z, y, x, H, W) (((int64_t)(z) * (H) + (y)) * (W) + (x))

/* ---- neighbourhood attention, CUDA-operator contract ------------------------------------------------------------
 * q (B,heads,D,H,W,d) pre-scaled, kpad (B,heads,D+2,H+2,W+2,d) zero padded, rpb (heads,27) or NULL,
 * attn (B,heads,D,H,W,27), token y = 9*ki+3*kj+kk.   ModeT-cu/modet/modet_kernel.cu:17-87 (forward),
 * :156-207 (dq), :209-267 + include/utils.h:29-38 (dk over the padded volume), :269-317 (drpb). */
void ref_modet_fw(const double* q, const double* kpad, const double* rpb, double* attn, int B, int heads, int D, int H,
                  int W, int d) {
  const int Hp = H + 2, Wp = W + 2;
  const int64_t V = (int64_t)D * H * W, Vp = (int64_t)(D + 2) * Hp * Wp;
  for (int bh = 0; bh < B * heads; ++bh)
    for (int i = 0; i < D; ++i)
      for (int j = 0; j < H; ++j)
        for (int k = 0; k < W; ++k) {
          const double* qp = q + ((int64_t)bh * V + IDX3(i, j, k, H, W)) * d;
          double* ap = attn + ((int64_t)bh * V + IDX3(i, j, k, H, W)) * 27;
          for (int ki = 0; ki < 3; ++ki)
            for (int kj = 0; kj < 3; ++kj)
              for (int kk = 0; kk < 3; ++kk) {
                const double* kp = kpad + ((int64_t)bh * Vp + IDX3(i + ki, j + kj, k + kk, Hp, Wp)) * d;
                double s = 0.0;
                for (int c = 0; c < d; ++c) s += qp[c] * kp[c];
                const int y = ki * 9 + kj * 3 + kk;
                ap[y] = s + (rpb ? rpb[(bh % heads) * 27 + y] : 0.0);
              }
        }
}

void ref_modet_bw(const double* d_attn, const double* q, const double* kpad, double* dq, double* dkpad, double* drpb,
                  int B, int heads, int D, int H, int W, int d) {
  const int Hp = H + 2, Wp = W + 2;
  const int64_t V = (int64_t)D * H * W, Vp = (int64_t)(D + 2) * Hp * Wp;
  memset(dq, 0, sizeof(double) * B * heads * V * d);
  memset(dkpad, 0, sizeof(double) * B * heads * Vp * d);
  if (drpb) memset(drpb, 0, sizeof(double) * heads * 27);
  for (int bh = 0; bh < B * heads; ++bh)
    for (int i = 0; i < D; ++i)
      for (int j = 0; j < H; ++j)
        for (int k = 0; k < W; ++k) {
          const int64_t n = (int64_t)bh * V + IDX3(i, j, k, H, W);
          for (int ki = 0; ki < 3; ++ki)
            for (int kj = 0; kj < 3; ++kj)
              for (int kk = 0; kk < 3; ++kk) {
                const int y = ki * 9 + kj * 3 + kk;
                const double g = d_attn[n * 27 + y];
                const int64_t m = (int64_t)bh * Vp + IDX3(i + ki, j + kj, k + kk, Hp, Wp);
                for (int c = 0; c < d; ++c) {
                  dq[n * d + c] += g * kpad[m * d + c];       /* :198-203 */
                  dkpad[m * d + c] += g * q[n * d + c];       /* scatter form of the gather at :250-262 */
                }
                if (drpb) drpb[(bh % heads) * 27 + y] += g;   /* :307-315 */
              }
        }
}

/* ---- fused ModeTransformer.forward: channels-last q,k (B,D,H,W,heads*d) -> out (B,D,H,W,heads*3) -----------------
 * ModeT/models.py:308-334: logits = scale*q.k(n+off)+rpb (zero key outside the volume), softmax over 27, sum p*off. */
void ref_na_fwd(const double* q, const double* k, const double* rpb, double* out, int B, int D, int H, int W, int heads,
                int d, double scale) {
  const int C = heads * d;
  for (int b = 0; b < B; ++b)
    for (int z = 0; z < D; ++z)
      for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x)
          for (int h = 0; h < heads; ++h) {
            const int64_t n = (int64_t)b * D * H * W + IDX3(z, y, x, H, W);
            double lg[27], m = -1e300;
            for (int t = 0; t < 27; ++t) {
              const int zz = z + t / 9 - 1, yy = y + (t / 3) % 3 - 1, xx = x + t % 3 - 1;
              double s = 0.0;
              if (zz >= 0 && zz < D && yy >= 0 && yy < H && xx >= 0 && xx < W) {
                const int64_t mm = (int64_t)b * D * H * W + IDX3(zz, yy, xx, H, W);
                for (int c = 0; c < d; ++c) s += scale * q[n * C + h * d + c] * k[mm * C + h * d + c];
              }
              lg[t] = s + rpb[h * 27 + t];
              if (lg[t] > m) m = lg[t];
            }
            double sum = 0.0, o[3] = {0, 0, 0};
            for (int t = 0; t < 27; ++t) { lg[t] = exp(lg[t] - m); sum += lg[t]; }
            for (int t = 0; t < 27; ++t) {
              o[0] += lg[t] * (t / 9 - 1); o[1] += lg[t] * ((t / 3) % 3 - 1); o[2] += lg[t] * (t % 3 - 1);
            }
            for (int a = 0; a < 3; ++a) out[n * heads * 3 + h * 3 + a] = o[a] / sum;
          }
}

/* ---- SpatialTransformer (ModeT/models.py:25-67): NCDHW src (B,C,D,H,W), flow (B,3,D,H,W), zero padding ----------
 * mode 0: trilinear at p+flow (grid_sample(align_corners=True) after the normalise round trip == voxel coordinates);
 * mode 1: nearest = nearbyint (round half to even), as ATen's grid_sampler. */
void ref_warp(const double* src, const double* flow, double* out, int B, int C, int D, int H, int W, int mode) {
  const int64_t V = (int64_t)D * H * W;
  for (int b = 0; b < B; ++b)
    for (int z = 0; z < D; ++z)
      for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
          const int64_t p = IDX3(z, y, x, H, W);
          const double sz = z + flow[((int64_t)b * 3 + 0) * V + p], sy = y + flow[((int64_t)b * 3 + 1) * V + p],
                       sx = x + flow[((int64_t)b * 3 + 2) * V + p];
          for (int c = 0; c < C; ++c) {
            const double* s = src + ((int64_t)b * C + c) * V;
            double acc = 0.0;
            if (mode == 1) {
              const double rz = nearbyint(sz), ry = nearbyint(sy), rx = nearbyint(sx);
              if (rz >= 0 && rz < D && ry >= 0 && ry < H && rx >= 0 && rx < W) acc = s[IDX3((int)rz, (int)ry, (int)rx, H, W)];
            } else {
              const double fz0 = floor(sz), fy0 = floor(sy), fx0 = floor(sx);
              for (int dz = 0; dz < 2; ++dz)
                for (int dy = 0; dy < 2; ++dy)
                  for (int dx = 0; dx < 2; ++dx) {
                    const double zz = fz0 + dz, yy = fy0 + dy, xx = fx0 + dx;
                    if (zz < 0 || zz >= D || yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
                    const double w = (dz ? sz - fz0 : 1 - (sz - fz0)) * (dy ? sy - fy0 : 1 - (sy - fy0)) *
                                     (dx ? sx - fx0 : 1 - (sx - fx0));
                    acc += w * s[IDX3((int)zz, (int)yy, (int)xx, H, W)];
                  }
            }
            out[((int64_t)b * C + c) * V + p] = acc;
          }
        }
}

/* ---- ConvBlock / ConvInsBlock (ModeT/models.py:119-151): conv3d(3,1,1)+bias [+InstanceNorm3d eps 1e-5] + LReLU(0.1) */
void ref_conv_block(const double* x, const double* w, const double* bias, double* raw, double* out, int B, int Cin,
                    int Cout, int D, int H, int W, int inst_norm) {
  const int64_t V = (int64_t)D * H * W;
  for (int b = 0; b < B; ++b)
    for (int co = 0; co < Cout; ++co) {
      double* r = raw + ((int64_t)b * Cout + co) * V;
      for (int z = 0; z < D; ++z)
        for (int y = 0; y < H; ++y)
          for (int xx = 0; xx < W; ++xx) {
            double s = bias[co];
            for (int ci = 0; ci < Cin; ++ci)
              for (int t = 0; t < 27; ++t) {
                const int zz = z + t / 9 - 1, yy = y + (t / 3) % 3 - 1, x2 = xx + t % 3 - 1;
                if (zz < 0 || zz >= D || yy < 0 || yy >= H || x2 < 0 || x2 >= W) continue;
                s += w[((int64_t)co * Cin + ci) * 27 + t] * x[((int64_t)b * Cin + ci) * V + IDX3(zz, yy, x2, H, W)];
              }
            r[IDX3(z, y, xx, H, W)] = s;
          }
      double mean = 0.0, var = 0.0;
      if (inst_norm) {
        for (int64_t i = 0; i < V; ++i) mean += r[i];
        mean /= (double)V;
        for (int64_t i = 0; i < V; ++i) var += (r[i] - mean) * (r[i] - mean);
        var /= (double)V;                                  /* biased variance */
      }
      double* o = out + ((int64_t)b * Cout + co) * V;
      for (int64_t i = 0; i < V; ++i) {
        const double v = inst_norm ? (r[i] - mean) / sqrt(var + 1e-5) : r[i];
        o[i] = v > 0 ? v : 0.1 * v;
      }
    }
}

/* ---- NCC_vxm (ModeT/losses.py:34-94): direct zero-padded 9^3 box sums, -mean(cc) ---------------------------------- */
double ref_ncc(const double* I, const double* J, int B, int D, int H, int W) {
  const int64_t V = (int64_t)D * H * W;
  double total = 0.0;
  for (int b = 0; b < B; ++b)
    for (int z = 0; z < D; ++z)
      for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
          double si = 0, sj = 0, sii = 0, sjj = 0, sij = 0;
          for (int dz = -4; dz <= 4; ++dz)
            for (int dy = -4; dy <= 4; ++dy)
              for (int dx = -4; dx <= 4; ++dx) {
                const int zz = z + dz, yy = y + dy, xx = x + dx;
                if (zz < 0 || zz >= D || yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
                const double a = I[b * V + IDX3(zz, yy, xx, H, W)], c = J[b * V + IDX3(zz, yy, xx, H, W)];
                si += a; sj += c; sii += a * a; sjj += c * c; sij += a * c;
              }
          const double n = 729.0, ui = si / n, uj = sj / n;
          const double cross = sij - uj * si - ui * sj + ui * uj * n;           /* losses.py:89 */
          const double iv = sii - 2 * ui * si + ui * ui * n, jv = sjj - 2 * uj * sj + uj * uj * n;
          total += cross * cross / (iv * jv + 1e-5);
        }
  return -total / (double)(B * V);
}

/* ---- Grad3d('l2') (ModeT/losses.py:6-31) on (B,3,D,H,W) ----------------------------------------------------------- */
double ref_grad3d(const double* f, int B, int D, int H, int W) {
  const int64_t V = (int64_t)D * H * W;
  double sd = 0, sh = 0, sw = 0;
  for (int bc = 0; bc < B * 3; ++bc)
    for (int z = 0; z < D; ++z)
      for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
          const double v = f[bc * V + IDX3(z, y, x, H, W)];
          if (z + 1 < D) { const double t = f[bc * V + IDX3(z + 1, y, x, H, W)] - v; sd += t * t; }
          if (y + 1 < H) { const double t = f[bc * V + IDX3(z, y + 1, x, H, W)] - v; sh += t * t; }
          if (x + 1 < W) { const double t = f[bc * V + IDX3(z, y, x + 1, H, W)] - v; sw += t * t; }
        }
  const double nb = (double)B * 3;
  return (sh / (nb * D * (H - 1) * W) + sd / (nb * (D - 1) * H * W) + sw / (nb * D * H * (W - 1))) / 3.0;
}

/* ---- dice_val_VOI (ModeT/utils.py:86-106): mean over labels 1..nlabels of 2|A&B| / (|A|+|B|+1e-5) ------------------ */
double ref_dice(const int16_t* pred, const int16_t* truth, int64_t n, int nlabels) {
  double tot = 0.0;
  for (int l = 1; l <= nlabels; ++l) {
    double a = 0, b = 0, ab = 0;
    for (int64_t i = 0; i < n; ++i) {
      const int p = pred[i] == l, t = truth[i] == l;
      a += p; b += t; ab += p & t;
    }
    tot += 2.0 * ab / (a + b + 1e-5);
  }
  return tot / nlabels;
}
