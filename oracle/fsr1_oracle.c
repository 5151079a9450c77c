/* TEST INFRASTRUCTURE — not product code.
 *
 * CPU restatement ("port") of the FSR 1.0 hot path in plain C11.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this; the product library
 * (fidelityfx-fsr_amd/csrc) never does.
 *
 * Parity pin: every entry point here is checked bit-for-bit (fp32) against oracle/_ref — the
 * reference's own ffx_a.h / ffx_fsr1.h compiled verbatim (oracle/build_ref.sh) — and against the
 * committed vectors under tests/golden/ generated from that build (tests/test_oracle.py).  The
 * reference repository itself ships no tests or golden vectors.
 *
 * Pinned semantics (the shading languages leave these open; same choices as oracle/_ref):
 *   - no FMA contraction: build with -ffp-contract=off; expressions keep the reference's order
 *   - min/max are IEEE minNum/maxNum (a NaN operand loses) with -0 < +0, as v_min_f32 / v_max_f32 and their _f16 forms
 *     evaluate them on the GPUs the reference's shaders run on; C's fminf / fmaxf may return either zero
 *   - ARcpF1 is the correctly rounded 1.0f/x
 *   - EASU taps are clamped to the resource edge; RCAS loads outside the image return 0
 *   - "H" (16-bit) arithmetic: each operation rounded once to binary16, round-to-nearest-even,
 *     subnormals kept; ARcpH = correctly rounded 1/x
 *
 * Image layout: RGBA interleaved, 4 floats per pixel, row-major, no padding.  The H entry
 * points take and return floats whose values are binary16-representable.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

enum { ORACLE_RCAS_DENOISE = 1, ORACLE_RCAS_ALPHA = 2, ORACLE_HDR_SQUARE = 4 };

static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline float omin(float a, float b) { return (a == 0.0f && b == 0.0f) ? (signbit(a) ? a : b) : fminf(a, b); }  /* -0 < +0 */
static inline float omax(float a, float b) { return (a == 0.0f && b == 0.0f) ? (signbit(a) ? b : a) : fmaxf(a, b); }

/* ------------------------------------------------------------------------------------------ */
/* constant setup                                                                             */
/* ------------------------------------------------------------------------------------------ */

/* ffx_a.h:482-549 AU1_AH1_AF1 — float -> half by *truncation* (table driven in the reference),
 * +-inf/NaN/overflow -> +-65504 (0x7bff), below 2^-25 -> 0.  Restated arithmetically: the tables
 * are base[i] / shift[i] indexed by sign+exponent i = u>>23. */
uint32_t oracle_AU1_AH1_AF1(float f) {
  uint32_t u = f2u(f);
  uint32_t sign = (u >> 16) & 0x8000u;
  uint32_t e = (u >> 23) & 0xffu;
  uint32_t m = u & 0x7fffffu;
  if (e < 103u) return sign;                          /* base 0, shift 24 */
  if (e < 113u) return sign + (0x0400u >> (113u - e)) + (m >> (126u - e)); /* subnormal half */
  if (e < 143u) return sign + ((e - 112u) << 10) + (m >> 13);             /* normal half */
  return sign + 0x7bffu;                              /* too large, inf, NaN */
}

/* ffx_fsr1.h:156-202 FsrEasuCon.  con16 = con0[4] con1[4] con2[4] con3[4]. */
void oracle_FsrEasuCon(uint32_t* con16, float vpX, float vpY, float inX, float inY, float outX, float outY) {
  uint32_t* con0 = con16; uint32_t* con1 = con16 + 4; uint32_t* con2 = con16 + 8; uint32_t* con3 = con16 + 12;
  con0[0] = f2u(vpX * (1.0f / outX));                     /* :171 */
  con0[1] = f2u(vpY * (1.0f / outY));                     /* :172 */
  con0[2] = f2u(0.5f * vpX * (1.0f / outX) - 0.5f);       /* :173 */
  con0[3] = f2u(0.5f * vpY * (1.0f / outY) - 0.5f);       /* :174 */
  con1[0] = f2u(1.0f / inX);                              /* :177 */
  con1[1] = f2u(1.0f / inY);                              /* :178 */
  con1[2] = f2u(1.0f * (1.0f / inX));                     /* :193 */
  con1[3] = f2u(-1.0f * (1.0f / inY));                    /* :194 */
  con2[0] = f2u(-1.0f * (1.0f / inX));                    /* :196 */
  con2[1] = f2u(2.0f * (1.0f / inY));                     /* :197 */
  con2[2] = f2u(1.0f * (1.0f / inX));                     /* :198 */
  con2[3] = f2u(2.0f * (1.0f / inY));                     /* :199 */
  con3[0] = f2u(0.0f * (1.0f / inX));                     /* :200 */
  con3[1] = f2u(4.0f * (1.0f / inY));                     /* :201 */
  con3[2] = con3[3] = 0;                                  /* :202 */
}

/* ffx_fsr1.h:205-225 FsrEasuConOffset. */
void oracle_FsrEasuConOffset(uint32_t* con16, float vpX, float vpY, float inX, float inY, float outX, float outY,
                             float offX, float offY) {
  oracle_FsrEasuCon(con16, vpX, vpY, inX, inY, outX, outY);
  con16[2] = f2u(0.5f * vpX * (1.0f / outX) - 0.5f + offX); /* :223 */
  con16[3] = f2u(0.5f * vpY * (1.0f / outY) - 0.5f + offY); /* :224 */
}

/* ffx_fsr1.h:662-672 FsrRcasCon (exp2f from libm, ffx_a.h:283/286). */
void oracle_FsrRcasCon(uint32_t* con, float sharpness) {
  sharpness = exp2f(-sharpness);
  con[0] = f2u(sharpness);
  con[1] = oracle_AU1_AH1_AF1(sharpness) + (oracle_AU1_AH1_AF1(sharpness) << 16); /* ffx_a.h:552 */
  con[2] = 0;
  con[3] = 0;
}

/* ------------------------------------------------------------------------------------------ */
/* fp32 helpers (ffx_a.h)                                                                     */
/* ------------------------------------------------------------------------------------------ */
static inline float APrxLoRcpF1(float a) { return u2f(0x7ef07ebbu - f2u(a)); }                 /* :1843 */
static inline float APrxMedRcpF1(float a) { float b = u2f(0x7ef19fffu - f2u(a)); return b * (-b * a + 2.0f); } /* :1844 */
static inline float APrxLoRsqF1(float a) { return u2f(0x5f347d74u - (f2u(a) >> 1)); }          /* :1845 */
static inline float ASatF1(float x) { return omin(omax(x, 0.0f), 1.0f); }                    /* :747 clamp(x,0,1) */
static inline float AMin3F1(float x, float y, float z) { return omin(x, omin(y, z)); }        /* :703 */
static inline float AMax3F1(float x, float y, float z) { return omax(x, omax(y, z)); }        /* :675 */
static inline float ARcpF1(float x) { return 1.0f / x; }                                        /* :737 */

typedef struct { const float* p; int w, h; } image_t;

/* Texel fetch with clamp-to-edge.  The reference reaches its 12 taps through four gather4 calls
 * at normalized coordinates p0..p3 (ffx_fsr1.h:344-360; callback FSR_Pass.hlsl:39-41); with
 * i=floor(p.x*W-0.5), j=floor(p.y*H-0.5) those resolve to integer texels (fp.x+dx, fp.y+dy),
 * dx,dy in [-1,2], each coordinate clamped by the CLAMP sampler. */
static inline const float* texel(const image_t* s, int x, int y) {
  x = x < 0 ? 0 : (x >= s->w ? s->w - 1 : x);
  y = y < 0 ? 0 : (y >= s->h ? s->h - 1 : y);
  return s->p + ((size_t)y * s->w + x) * 4;
}

/* ffx_fsr1.h:275-313 FsrEasuSetF.  w is the bilinear weight selected by the biS/T/U/V predicate. */
static inline void FsrEasuSetF(float dir[2], float* len, float w, float lA, float lB, float lC, float lD, float lE) {
  float dc = lD - lC;
  float cb = lC - lB;
  float lenX = omax(fabsf(dc), fabsf(cb));
  lenX = APrxLoRcpF1(lenX);
  float dirX = lD - lB;
  dir[0] += dirX * w;
  lenX = ASatF1(fabsf(dirX) * lenX);
  lenX *= lenX;
  *len += lenX * w;
  float ec = lE - lC;
  float ca = lC - lA;
  float lenY = omax(fabsf(ec), fabsf(ca));
  lenY = APrxLoRcpF1(lenY);
  float dirY = lE - lA;
  dir[1] += dirY * w;
  lenY = ASatF1(fabsf(dirY) * lenY);
  lenY *= lenY;
  *len += lenY * w;
}

/* ffx_fsr1.h:239-272 FsrEasuTapF. */
static inline void FsrEasuTapF(float aC[3], float* aW, float offX, float offY, const float dir[2], const float len[2],
                               float lob, float clp, const float* c) {
  float vx = (offX * (dir[0])) + (offY * dir[1]);
  float vy = (offX * (-dir[1])) + (offY * dir[0]);
  vx *= len[0];
  vy *= len[1];
  float d2 = vx * vx + vy * vy;
  d2 = omin(d2, clp);
  float wB = (float)(2.0 / 5.0) * d2 + -1.0f;
  float wA = lob * d2 + -1.0f;
  wB *= wB;
  wA *= wA;
  wB = (float)(25.0 / 16.0) * wB + (float)(-(25.0 / 16.0 - 1.0));
  float w = wB * wA;
  aC[0] += c[0] * w;
  aC[1] += c[1] * w;
  aC[2] += c[2] * w;
  *aW += w;
}

/* ffx_fsr1.h:315-437 FsrEasuF. */
static void FsrEasuF(float pix[3], uint32_t ipx, uint32_t ipy, const uint32_t* con, const image_t* src) {
  /* :324-326 */
  float ppx = (float)ipx * u2f(con[0]) + u2f(con[2]);
  float ppy = (float)ipy * u2f(con[1]) + u2f(con[3]);
  float fpx = floorf(ppx), fpy = floorf(ppy);
  ppx -= fpx;
  ppy -= fpy;
  int fx = (int)fpx, fy = (int)fpy;
  /* :328-360 12 taps   b c / e f g h / i j k l / n o */
  const float* b = texel(src, fx, fy - 1);     const float* c = texel(src, fx + 1, fy - 1);
  const float* e = texel(src, fx - 1, fy);     const float* f = texel(src, fx, fy);
  const float* g = texel(src, fx + 1, fy);     const float* h = texel(src, fx + 2, fy);
  const float* i = texel(src, fx - 1, fy + 1); const float* j = texel(src, fx, fy + 1);
  const float* k = texel(src, fx + 1, fy + 1); const float* l = texel(src, fx + 2, fy + 1);
  const float* n = texel(src, fx, fy + 2);     const float* o = texel(src, fx + 1, fy + 2);
  /* :363-366 luma*2 = B*0.5 + (R*0.5 + G) */
#define LUMA(t) ((t)[2] * 0.5f + ((t)[0] * 0.5f + (t)[1]))
  float bL = LUMA(b), cL = LUMA(c), eL = LUMA(e), fL = LUMA(f), gL = LUMA(g), hL = LUMA(h);
  float iL = LUMA(i), jL = LUMA(j), kL = LUMA(k), lL = LUMA(l), nL = LUMA(n), oL = LUMA(o);
#undef LUMA
  /* :381-386 with the weights of :284-288 */
  float dir[2] = {0.0f, 0.0f};
  float len = 0.0f;
  FsrEasuSetF(dir, &len, (1.0f - ppx) * (1.0f - ppy), bL, eL, fL, gL, jL);
  FsrEasuSetF(dir, &len, ppx * (1.0f - ppy), cL, fL, gL, hL, kL);
  FsrEasuSetF(dir, &len, (1.0f - ppx) * ppy, fL, iL, jL, kL, nL);
  FsrEasuSetF(dir, &len, ppx * ppy, gL, jL, kL, lL, oL);
  /* :389-395 */
  float dir2x = dir[0] * dir[0], dir2y = dir[1] * dir[1];
  float dirR = dir2x + dir2y;
  int zro = dirR < (float)(1.0 / 32768.0);
  dirR = APrxLoRsqF1(dirR);
  dirR = zro ? 1.0f : dirR;
  dir[0] = zro ? 1.0f : dir[0];
  dir[0] *= dirR;
  dir[1] *= dirR;
  /* :397-409 */
  len = len * 0.5f;
  len *= len;
  float stretch = (dir[0] * dir[0] + dir[1] * dir[1]) * APrxLoRcpF1(omax(fabsf(dir[0]), fabsf(dir[1])));
  float len2[2] = {1.0f + (stretch - 1.0f) * len, 1.0f + -0.5f * len};
  float lob = 0.5f + (float)((1.0 / 4.0 - 0.04) - 0.5) * len;
  float clp = APrxLoRcpF1(lob);
  /* :416-419 min/max of the 4 nearest (f,g,j,k) */
  float min4[3], max4[3];
  for (int ch = 0; ch < 3; ++ch) {
    min4[ch] = omin(AMin3F1(f[ch], g[ch], j[ch]), k[ch]);
    max4[ch] = omax(AMax3F1(f[ch], g[ch], j[ch]), k[ch]);
  }
  /* :421-434 accumulation, reference order b c i j f e k l h g o n */
  float aC[3] = {0.0f, 0.0f, 0.0f};
  float aW = 0.0f;
  FsrEasuTapF(aC, &aW, 0.0f - ppx, -1.0f - ppy, dir, len2, lob, clp, b);
  FsrEasuTapF(aC, &aW, 1.0f - ppx, -1.0f - ppy, dir, len2, lob, clp, c);
  FsrEasuTapF(aC, &aW, -1.0f - ppx, 1.0f - ppy, dir, len2, lob, clp, i);
  FsrEasuTapF(aC, &aW, 0.0f - ppx, 1.0f - ppy, dir, len2, lob, clp, j);
  FsrEasuTapF(aC, &aW, 0.0f - ppx, 0.0f - ppy, dir, len2, lob, clp, f);
  FsrEasuTapF(aC, &aW, -1.0f - ppx, 0.0f - ppy, dir, len2, lob, clp, e);
  FsrEasuTapF(aC, &aW, 1.0f - ppx, 1.0f - ppy, dir, len2, lob, clp, k);
  FsrEasuTapF(aC, &aW, 2.0f - ppx, 1.0f - ppy, dir, len2, lob, clp, l);
  FsrEasuTapF(aC, &aW, 2.0f - ppx, 0.0f - ppy, dir, len2, lob, clp, h);
  FsrEasuTapF(aC, &aW, 1.0f - ppx, 0.0f - ppy, dir, len2, lob, clp, g);
  FsrEasuTapF(aC, &aW, 1.0f - ppx, 2.0f - ppy, dir, len2, lob, clp, o);
  FsrEasuTapF(aC, &aW, 0.0f - ppx, 2.0f - ppy, dir, len2, lob, clp, n);
  /* :437 normalize and dering */
  float rW = ARcpF1(aW);
  for (int ch = 0; ch < 3; ++ch) pix[ch] = omin(max4[ch], omax(min4[ch], aC[ch] * rW));
}

/* Rows [y0,y1) of EASU; alpha is written as 1 and `c*=c` applied when flags has HDR_SQUARE
 * (dispatch shell, sample/src/DX12/FSR_Pass.hlsl:75-80). */
void oracle_easu_f(const float* in, int inW, int inH, float* out, int outW, int outH, const uint32_t* con16,
                   int flags, int y0, int y1) {
  image_t src = {in, inW, inH};
  (void)outH;
#pragma omp parallel for schedule(dynamic, 4)
  for (int y = y0; y < y1; ++y)
    for (int x = 0; x < outW; ++x) {
      float c[3];
      FsrEasuF(c, (uint32_t)x, (uint32_t)y, con16, &src);
      if (flags & ORACLE_HDR_SQUARE) { c[0] *= c[0]; c[1] *= c[1]; c[2] *= c[2]; }
      float* o = out + ((size_t)y * outW + x) * 4;
      o[0] = c[0]; o[1] = c[1]; o[2] = c[2]; o[3] = 1.0f;
    }
}

/* ------------------------------------------------------------------------------------------ */
/* RCAS fp32                                                                                  */
/* ------------------------------------------------------------------------------------------ */
static inline void load0(const image_t* s, int x, int y, float t[4]) { /* FSR_Pass.hlsl:45 Load; OOB = 0 */
  if (x < 0 || y < 0 || x >= s->w || y >= s->h) { t[0] = t[1] = t[2] = t[3] = 0.0f; return; }
  const float* q = s->p + ((size_t)y * s->w + x) * 4;
  t[0] = q[0]; t[1] = q[1]; t[2] = q[2]; t[3] = q[3];
}

/* ffx_fsr1.h:684-769 FsrRcasF (FSR_RCAS_DENOISE :761-763 and FSR_RCAS_PASSTHROUGH_ALPHA :688-705 as flags). */
static void FsrRcasF(float pix[4], int x, int y, const uint32_t* con, const image_t* src, int flags) {
  float b[4], d[4], e[4], f[4], h[4];
  load0(src, x, y - 1, b);
  load0(src, x - 1, y, d);
  load0(src, x, y, e);
  load0(src, x + 1, y, f);
  load0(src, x, y + 1, h);
  float bR = b[0], bG = b[1], bB = b[2], dR = d[0], dG = d[1], dB = d[2], eR = e[0], eG = e[1], eB = e[2];
  float fR = f[0], fG = f[1], fB = f[2], hR = h[0], hG = h[1], hB = h[2];
  /* :731-735 */
  float bL = bB * 0.5f + (bR * 0.5f + bG);
  float dL = dB * 0.5f + (dR * 0.5f + dG);
  float eL = eB * 0.5f + (eR * 0.5f + eG);
  float fL = fB * 0.5f + (fR * 0.5f + fG);
  float hL = hB * 0.5f + (hR * 0.5f + hG);
  /* :737-739 */
  float nz = 0.25f * bL + 0.25f * dL + 0.25f * fL + 0.25f * hL - eL;
  nz = ASatF1(fabsf(nz) * APrxMedRcpF1(AMax3F1(AMax3F1(bL, dL, eL), fL, hL) - AMin3F1(AMin3F1(bL, dL, eL), fL, hL)));
  nz = -0.5f * nz + 1.0f;
  /* :741-746 */
  float mn4R = omin(AMin3F1(bR, dR, fR), hR);
  float mn4G = omin(AMin3F1(bG, dG, fG), hG);
  float mn4B = omin(AMin3F1(bB, dB, fB), hB);
  float mx4R = omax(AMax3F1(bR, dR, fR), hR);
  float mx4G = omax(AMax3F1(bG, dG, fG), hG);
  float mx4B = omax(AMax3F1(bB, dB, fB), hB);
  /* :748-758 */
  const float peakCx = 1.0f, peakCy = -1.0f * 4.0f;
  float hitMinR = omin(mn4R, eR) * ARcpF1(4.0f * mx4R);
  float hitMinG = omin(mn4G, eG) * ARcpF1(4.0f * mx4G);
  float hitMinB = omin(mn4B, eB) * ARcpF1(4.0f * mx4B);
  float hitMaxR = (peakCx - omax(mx4R, eR)) * ARcpF1(4.0f * mn4R + peakCy);
  float hitMaxG = (peakCx - omax(mx4G, eG)) * ARcpF1(4.0f * mn4G + peakCy);
  float hitMaxB = (peakCx - omax(mx4B, eB)) * ARcpF1(4.0f * mn4B + peakCy);
  float lobeR = omax(-hitMinR, hitMaxR);
  float lobeG = omax(-hitMinG, hitMaxG);
  float lobeB = omax(-hitMinB, hitMaxB);
  /* :759 FSR_RCAS_LIMIT = 0.25-1/16 (:654) */
  float lobe = omax((float)(-(0.25 - (1.0 / 16.0))), omin(AMax3F1(lobeR, lobeG, lobeB), 0.0f)) * u2f(con[0]);
  if (flags & ORACLE_RCAS_DENOISE) lobe *= nz;
  /* :765-768 */
  float rcpL = APrxMedRcpF1(4.0f * lobe + 1.0f);
  pix[0] = (lobe * bR + lobe * dR + lobe * hR + lobe * fR + eR) * rcpL;
  pix[1] = (lobe * bG + lobe * dG + lobe * hG + lobe * fG + eG) * rcpL;
  pix[2] = (lobe * bB + lobe * dB + lobe * hB + lobe * fB + eB) * rcpL;
  pix[3] = (flags & ORACLE_RCAS_ALPHA) ? e[3] : 1.0f;
}

void oracle_rcas_f(const float* in, int W, int H, float* out, const uint32_t* con, int flags, int y0, int y1) {
  image_t src = {in, W, H};
#pragma omp parallel for schedule(dynamic, 4)
  for (int y = y0; y < y1; ++y)
    for (int x = 0; x < W; ++x) {
      float c[4];
      FsrRcasF(c, x, y, con, &src, flags);
      if (flags & ORACLE_HDR_SQUARE) { c[0] *= c[0]; c[1] *= c[1]; c[2] *= c[2]; } /* FSR_Pass.hlsl:92-93 */
      float* o = out + ((size_t)y * W + x) * 4;
      o[0] = c[0]; o[1] = c[1]; o[2] = c[2]; o[3] = c[3];
    }
}

/* ------------------------------------------------------------------------------------------ */
/* binary16 emulation: values live in float, every op rounds once (RTNE) through double        */
/* ------------------------------------------------------------------------------------------ */
typedef float hf; /* a float that holds a binary16-representable value */

static inline hf hround(double v) {
  if (isnan(v) || isinf(v) || v == 0.0) return (float)v;
  double a = fabs(v);
  if (a >= 65520.0) return v < 0 ? -INFINITY : INFINITY;
  int e;
  frexp(a, &e);
  int q = e - 11;
  if (q < -24) q = -24;
  double r = ldexp(nearbyint(ldexp(a, -q)), q);
  return (float)(v < 0 ? -r : r);
}
static inline uint16_t hbits(hf v) {
  uint16_t sign = signbit(v) ? 0x8000u : 0u;
  if (isnan(v)) return (uint16_t)(sign | 0x7e00u);
  double a = fabs((double)v);
  if (isinf(a)) return (uint16_t)(sign | 0x7c00u);
  if (a == 0.0) return sign;
  int e;
  double m = frexp(a, &e);
  int E = e - 1;
  if (E < -14) return (uint16_t)(sign | (uint16_t)ldexp(a, 24));
  return (uint16_t)(sign | ((E + 15) << 10) | (uint16_t)ldexp(2.0 * m - 1.0, 10));
}
static inline hf hfrombits(uint16_t h) {
  int s = (h >> 15) & 1, e = (h >> 10) & 31, m = h & 1023;
  double v;
  if (e == 0) v = ldexp((double)m, -24);
  else if (e == 31) v = m ? NAN : INFINITY;
  else v = ldexp(1.0 + m / 1024.0, e - 15);
  return (float)(s ? -v : v);
}
static inline hf hadd(hf a, hf b) { return hround((double)a + (double)b); }
static inline hf hsub(hf a, hf b) { return hround((double)a - (double)b); }
static inline hf hmul(hf a, hf b) { return hround((double)a * (double)b); }
static inline hf hdiv(hf a, hf b) { return hround((double)a / (double)b); }
static inline hf hmin(hf a, hf b) { return omin(a, b); }
static inline hf hmax(hf a, hf b) { return omax(a, b); }
static inline hf hsat(hf a) { return hmin(hmax(a, 0.0f), 1.0f); }                     /* ffx_a.h:896 */
static inline hf APrxLoRcpH1(hf a) { return hfrombits((uint16_t)(0x7784u - hbits(a))); } /* ffx_a.h:1808 */
static inline hf APrxMedRcpH1(hf a) {                                                   /* ffx_a.h:1814 */
  hf b = hfrombits((uint16_t)(0x778du - hbits(a)));
  return hmul(b, hadd(hmul(-b, a), 2.0f));
}
static inline hf APrxLoRsqH1(hf a) { return hfrombits((uint16_t)(0x59a3u - (hbits(a) >> 1))); } /* ffx_a.h:1820 */

/* ffx_fsr1.h:476-503 FsrEasuSetH — two positions at once; arrays of 2 are the AH2 lanes. */
static inline void FsrEasuSetH(hf dirPX[2], hf dirPY[2], hf lenP[2], hf ppx, hf ppy, int biST, int biUV,
                               const hf lA[2], const hf lB[2], const hf lC[2], const hf lD[2], const hf lE[2]) {
  hf w[2] = {0.0f, 0.0f};
  /* :483-484  (AH2(1,0)+AH2(-pp.x,pp.x)) * AH2_(1-pp.y | pp.y) */
  hf wx0 = hadd(1.0f, -ppx), wx1 = hadd(0.0f, ppx);
  if (biST) { hf s = hsub(1.0f, ppy); w[0] = hmul(wx0, s); w[1] = hmul(wx1, s); }
  if (biUV) { w[0] = hmul(wx0, ppy); w[1] = hmul(wx1, ppy); }
  for (int t = 0; t < 2; ++t) {
    hf dc = hsub(lD[t], lC[t]);
    hf cb = hsub(lC[t], lB[t]);
    hf lenX = hmax(fabsf(dc), fabsf(cb));
    lenX = hdiv(1.0f, lenX);                       /* ARcpH2 :489 */
    hf dirX = hsub(lD[t], lB[t]);
    dirPX[t] = hadd(dirPX[t], hmul(dirX, w[t]));
    lenX = hsat(hmul(fabsf(dirX), lenX));
    lenX = hmul(lenX, lenX);
    lenP[t] = hadd(lenP[t], hmul(lenX, w[t]));
    hf ec = hsub(lE[t], lC[t]);
    hf ca = hsub(lC[t], lA[t]);
    hf lenY = hmax(fabsf(ec), fabsf(ca));
    lenY = hdiv(1.0f, lenY);                       /* ARcpH2 :498 */
    hf dirY = hsub(lE[t], lA[t]);
    dirPY[t] = hadd(dirPY[t], hmul(dirY, w[t]));
    lenY = hsat(hmul(fabsf(dirY), lenY));
    lenY = hmul(lenY, lenY);
    lenP[t] = hadd(lenP[t], hmul(lenY, w[t]));
  }
}

/* ffx_fsr1.h:452-473 FsrEasuTapH — two taps at once. */
static inline void FsrEasuTapH(hf aCR[2], hf aCG[2], hf aCB[2], hf aW[2], const hf offX[2], const hf offY[2],
                               const hf dir[2], const hf len[2], hf lob, hf clp, const hf cR[2], const hf cG[2],
                               const hf cB[2]) {
  for (int t = 0; t < 2; ++t) {
    hf vX = hadd(hmul(offX[t], dir[0]), hmul(offY[t], dir[1]));
    hf vY = hadd(hmul(offX[t], -dir[1]), hmul(offY[t], dir[0]));
    vX = hmul(vX, len[0]);
    vY = hmul(vY, len[1]);
    hf d2 = hadd(hmul(vX, vX), hmul(vY, vY));
    d2 = hmin(d2, clp);
    hf wB = hadd(hmul(hround(2.0 / 5.0), d2), -1.0f);
    hf wA = hadd(hmul(lob, d2), -1.0f);
    wB = hmul(wB, wB);
    wA = hmul(wA, wA);
    wB = hadd(hmul(hround(25.0 / 16.0), wB), hround(-(25.0 / 16.0 - 1.0)));
    hf w = hmul(wB, wA);
    aCR[t] = hadd(aCR[t], hmul(cR[t], w));
    aCG[t] = hadd(aCG[t], hmul(cG[t], w));
    aCB[t] = hadd(aCB[t], hmul(cB[t], w));
    aW[t] = hadd(aW[t], w);
  }
}

/* ffx_fsr1.h:505-593 FsrEasuH. */
static void FsrEasuH(hf pix[3], uint32_t ipx, uint32_t ipy, const uint32_t* con, const image_t* src) {
  /* :513-516 position in fp32, then ppp = AH2(pp) */
  float ppx = (float)ipx * u2f(con[0]) + u2f(con[2]);
  float ppy = (float)ipy * u2f(con[1]) + u2f(con[3]);
  float fpx = floorf(ppx), fpy = floorf(ppy);
  ppx -= fpx;
  ppy -= fpy;
  hf px = hround(ppx), py = hround(ppy);
  int fx = (int)fpx, fy = (int)fpy;
  const float* tb = texel(src, fx, fy - 1);     const float* tc = texel(src, fx + 1, fy - 1);
  const float* te = texel(src, fx - 1, fy);     const float* tf = texel(src, fx, fy);
  const float* tg = texel(src, fx + 1, fy);     const float* th = texel(src, fx + 2, fy);
  const float* ti = texel(src, fx - 1, fy + 1); const float* tj = texel(src, fx, fy + 1);
  const float* tk = texel(src, fx + 1, fy + 1); const float* tl = texel(src, fx + 2, fy + 1);
  const float* tn = texel(src, fx, fy + 2);     const float* to = texel(src, fx + 1, fy + 2);
  /* :535-538 */
#define LUMAH(t) hadd(hmul(hround((t)[2]), 0.5f), hadd(hmul(hround((t)[0]), 0.5f), hround((t)[1])))
  hf bL = LUMAH(tb), cL = LUMAH(tc), eL = LUMAH(te), fL = LUMAH(tf), gL = LUMAH(tg), hL = LUMAH(th);
  hf iL = LUMAH(ti), jL = LUMAH(tj), kL = LUMAH(tk), lL = LUMAH(tl), nL = LUMAH(tn), oL = LUMAH(to);
#undef LUMAH
  /* :552-558 */
  hf dirPX[2] = {0, 0}, dirPY[2] = {0, 0}, lenP[2] = {0, 0};
  {
    hf A[2] = {bL, cL}, B[2] = {eL, fL}, C[2] = {fL, gL}, D[2] = {gL, hL}, E[2] = {jL, kL};
    FsrEasuSetH(dirPX, dirPY, lenP, px, py, 1, 0, A, B, C, D, E);
  }
  {
    hf A[2] = {fL, gL}, B[2] = {iL, jL}, C[2] = {jL, kL}, D[2] = {kL, lL}, E[2] = {nL, oL};
    FsrEasuSetH(dirPX, dirPY, lenP, px, py, 0, 1, A, B, C, D, E);
  }
  hf dir[2] = {hadd(dirPX[0], dirPX[1]), hadd(dirPY[0], dirPY[1])};
  hf len = hadd(lenP[0], lenP[1]);
  /* :560-572 */
  hf dir2x = hmul(dir[0], dir[0]), dir2y = hmul(dir[1], dir[1]);
  hf dirR = hadd(dir2x, dir2y);
  int zro = dirR < hround(1.0 / 32768.0);
  dirR = APrxLoRsqH1(dirR);
  dirR = zro ? 1.0f : dirR;
  dir[0] = zro ? 1.0f : dir[0];
  dir[0] = hmul(dir[0], dirR);
  dir[1] = hmul(dir[1], dirR);
  len = hmul(len, 0.5f);
  len = hmul(len, len);
  hf stretch = hmul(hadd(hmul(dir[0], dir[0]), hmul(dir[1], dir[1])), APrxLoRcpH1(hmax(fabsf(dir[0]), fabsf(dir[1]))));
  hf len2[2] = {hadd(1.0f, hmul(hsub(stretch, 1.0f), len)), hadd(1.0f, hmul(-0.5f, len))};
  hf lob = hadd(0.5f, hmul(hround((1.0 / 4.0 - 0.04) - 0.5), len));
  hf clp = APrxLoRcpH1(lob);
  /* :575-577 min and max of f,g,j,k in one max() over (-x,x) pairs */
  hf both[3][2];
  for (int ch = 0; ch < 3; ++ch) {
    hf f = hround(tf[ch]), g = hround(tg[ch]), j = hround(tj[ch]), k = hround(tk[ch]);
    both[ch][0] = hmax(hmax(-f, -g), hmax(-j, -k));
    both[ch][1] = hmax(hmax(f, g), hmax(j, k));
  }
  /* :579-588 */
  hf pR[2] = {0, 0}, pG[2] = {0, 0}, pB[2] = {0, 0}, pW[2] = {0, 0};
#define TAPH(ox0, ox1, oy0, oy1, t0, t1)                                                          \
  {                                                                                                \
    hf offX[2] = {hsub((ox0), px), hsub((ox1), px)}, offY[2] = {hsub((oy0), py), hsub((oy1), py)}; \
    hf cR[2] = {hround((t0)[0]), hround((t1)[0])}, cG[2] = {hround((t0)[1]), hround((t1)[1])};      \
    hf cB[2] = {hround((t0)[2]), hround((t1)[2])};                                                  \
    FsrEasuTapH(pR, pG, pB, pW, offX, offY, dir, len2, lob, clp, cR, cG, cB);                       \
  }
  TAPH(0.0f, 1.0f, -1.0f, -1.0f, tb, tc)  /* bczz.xy  */
  TAPH(-1.0f, 0.0f, 1.0f, 1.0f, ti, tj)   /* ijfe.xy  */
  TAPH(0.0f, -1.0f, 0.0f, 0.0f, tf, te)   /* ijfe.zw  */
  TAPH(1.0f, 2.0f, 1.0f, 1.0f, tk, tl)    /* klhg.xy  */
  TAPH(2.0f, 1.0f, 0.0f, 0.0f, th, tg)    /* klhg.zw  */
  TAPH(1.0f, 0.0f, 2.0f, 2.0f, to, tn)    /* zzon.zw  */
#undef TAPH
  hf aC[3] = {hadd(pR[0], pR[1]), hadd(pG[0], pG[1]), hadd(pB[0], pB[1])};
  hf aW = hadd(pW[0], pW[1]);
  /* :593 */
  hf rW = hdiv(1.0f, aW);
  for (int ch = 0; ch < 3; ++ch) pix[ch] = hmin(both[ch][1], hmax(-both[ch][0], hmul(aC[ch], rW)));
}

void oracle_easu_h(const float* in, int inW, int inH, float* out, int outW, int outH, const uint32_t* con16,
                   int flags, int y0, int y1) {
  image_t src = {in, inW, inH};
  (void)outH;
#pragma omp parallel for schedule(dynamic, 4)
  for (int y = y0; y < y1; ++y)
    for (int x = 0; x < outW; ++x) {
      hf c[3];
      FsrEasuH(c, (uint32_t)x, (uint32_t)y, con16, &src);
      if (flags & ORACLE_HDR_SQUARE) { c[0] = hmul(c[0], c[0]); c[1] = hmul(c[1], c[1]); c[2] = hmul(c[2], c[2]); }
      float* o = out + ((size_t)y * outW + x) * 4;
      o[0] = c[0]; o[1] = c[1]; o[2] = c[2]; o[3] = 1.0f;
    }
}

/* ffx_fsr1.h:782-866 FsrRcasH (non-packed 16-bit form, the one the sample dispatches). */
static void FsrRcasH(hf pix[4], int x, int y, const uint32_t* con, const image_t* src, int flags) {
  float b[4], d[4], e[4], f[4], h[4];
  load0(src, x, y - 1, b);
  load0(src, x - 1, y, d);
  load0(src, x, y, e);
  load0(src, x + 1, y, f);
  load0(src, x, y + 1, h);
  hf bR = hround(b[0]), bG = hround(b[1]), bB = hround(b[2]);
  hf dR = hround(d[0]), dG = hround(d[1]), dB = hround(d[2]);
  hf eR = hround(e[0]), eG = hround(e[1]), eB = hround(e[2]);
  hf fR = hround(f[0]), fG = hround(f[1]), fB = hround(f[2]);
  hf hR = hround(h[0]), hG = hround(h[1]), hB = hround(h[2]);
#define LUMAH(B_, R_, G_) hadd(hmul(B_, 0.5f), hadd(hmul(R_, 0.5f), G_))
  hf bL = LUMAH(bB, bR, bG), dL = LUMAH(dB, dR, dG), eL = LUMAH(eB, eR, eG), fL = LUMAH(fB, fR, fG), hL = LUMAH(hB, hR, hG);
#undef LUMAH
  /* :835-837 */
  hf nz = hsub(hadd(hadd(hadd(hmul(0.25f, bL), hmul(0.25f, dL)), hmul(0.25f, fL)), hmul(0.25f, hL)), eL);
  hf rng = hsub(hmax(hmax(bL, hmax(dL, eL)), hmax(fL, hL)), hmin(hmin(bL, hmin(dL, eL)), hmin(fL, hL)));
  nz = hsat(hmul(fabsf(nz), APrxMedRcpH1(rng)));
  nz = hadd(hmul(-0.5f, nz), 1.0f);
  /* :839-844  AMin3H1(x,y,z)=min(x,min(y,z)) */
  hf mn4R = hmin(hmin(bR, hmin(dR, fR)), hR), mn4G = hmin(hmin(bG, hmin(dG, fG)), hG), mn4B = hmin(hmin(bB, hmin(dB, fB)), hB);
  hf mx4R = hmax(hmax(bR, hmax(dR, fR)), hR), mx4G = hmax(hmax(bG, hmax(dG, fG)), hG), mx4B = hmax(hmax(bB, hmax(dB, fB)), hB);
  /* :846-856 */
  const hf peakCx = 1.0f, peakCy = -4.0f;
  hf hitMinR = hmul(hmin(mn4R, eR), hdiv(1.0f, hmul(4.0f, mx4R)));
  hf hitMinG = hmul(hmin(mn4G, eG), hdiv(1.0f, hmul(4.0f, mx4G)));
  hf hitMinB = hmul(hmin(mn4B, eB), hdiv(1.0f, hmul(4.0f, mx4B)));
  hf hitMaxR = hmul(hsub(peakCx, hmax(mx4R, eR)), hdiv(1.0f, hadd(hmul(4.0f, mn4R), peakCy)));
  hf hitMaxG = hmul(hsub(peakCx, hmax(mx4G, eG)), hdiv(1.0f, hadd(hmul(4.0f, mn4G), peakCy)));
  hf hitMaxB = hmul(hsub(peakCx, hmax(mx4B, eB)), hdiv(1.0f, hadd(hmul(4.0f, mn4B), peakCy)));
  hf lobeR = hmax(-hitMinR, hitMaxR), lobeG = hmax(-hitMinG, hitMaxG), lobeB = hmax(-hitMinB, hitMaxB);
  /* :857 sharpness = AH2_AU1(con.y).x : the packed half of con[1] */
  hf sharp = hfrombits((uint16_t)(con[1] & 0xffffu));
  hf lobe = hmul(hmax(hround(-(0.25 - (1.0 / 16.0))), hmin(hmax(lobeR, hmax(lobeG, lobeB)), 0.0f)), sharp);
  if (flags & ORACLE_RCAS_DENOISE) lobe = hmul(lobe, nz);
  /* :863-866 */
  hf rcpL = APrxMedRcpH1(hadd(hmul(4.0f, lobe), 1.0f));
#define RESOLVE(b_, d_, h_, f_, e_) \
  hmul(hadd(hadd(hadd(hadd(hmul(lobe, b_), hmul(lobe, d_)), hmul(lobe, h_)), hmul(lobe, f_)), e_), rcpL)
  pix[0] = RESOLVE(bR, dR, hR, fR, eR);
  pix[1] = RESOLVE(bG, dG, hG, fG, eG);
  pix[2] = RESOLVE(bB, dB, hB, fB, eB);
#undef RESOLVE
  pix[3] = (flags & ORACLE_RCAS_ALPHA) ? hround(e[3]) : 1.0f;
}

void oracle_rcas_h(const float* in, int W, int H, float* out, const uint32_t* con, int flags, int y0, int y1) {
  image_t src = {in, W, H};
#pragma omp parallel for schedule(dynamic, 4)
  for (int y = y0; y < y1; ++y)
    for (int x = 0; x < W; ++x) {
      hf c[4];
      FsrRcasH(c, x, y, con, &src, flags);
      if (flags & ORACLE_HDR_SQUARE) { c[0] = hmul(c[0], c[0]); c[1] = hmul(c[1], c[1]); c[2] = hmul(c[2], c[2]); }
      float* o = out + ((size_t)y * W + x) * 4;
      o[0] = c[0]; o[1] = c[1]; o[2] = c[2]; o[3] = c[3];
    }
}

/* ffx_a.h:2304 ARmp8x8: lane -> (x,y) inside an 8x8 tile, 2x2 quads kept together. */
void oracle_ARmp8x8(uint32_t a, uint32_t* xy) {
  xy[0] = (a >> 1) & 7u;                         /* ABfe(a,1,3) */
  xy[1] = (((a >> 3) & 7u) & ~1u) | (a & 1u);    /* ABfiM(ABfe(a,3,3),a,1) */
}

/* round a float buffer to binary16 values in place (RTNE) — used by tests to model fp16 storage */
void oracle_round_to_half(float* p, size_t n) {
  for (size_t i = 0; i < n; ++i) p[i] = hround((double)p[i]);
}

int oracle_omp_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* ------------------------------------------------------------------------------------------ */
/* colour stages around the filters: LFGA, SRTM, TEPD (ffx_fsr1.h:986-1199), fp32              */
/* ------------------------------------------------------------------------------------------ */
enum { ORACLE_COLOR_SRTM = 1, ORACLE_COLOR_LFGA = 2, ORACLE_COLOR_SRTM_INV = 4, ORACLE_COLOR_TEPD_C8 = 8,
       ORACLE_COLOR_TEPD_C10 = 16, ORACLE_COLOR_DITHER_FROM_NOISE = 32 };

/* ffx_a.h:1499 AGtZeroF1 = ASatF1(m * +INF): 1 for m > 0, 0 for m < 0, and 0 for m == 0 (0*inf = NaN, which
 * the minNum/maxNum clamp turns into 0). */
static inline float AGtZeroF1(float m) { return ASatF1(m * INFINITY); }

/* :1042 FsrSrtmF   c *= rcp(max3(c) + 1) */
static inline void FsrSrtmF(float c[3]) {
  float r = ARcpF1(AMax3F1(c[0], c[1], c[2]) + 1.0f);
  c[0] *= r; c[1] *= r; c[2] *= r;
}
/* :1044 FsrSrtmInvF   c *= rcp(max(1/32768, 1 - max3(c))) */
static inline void FsrSrtmInvF(float c[3]) {
  float r = ARcpF1(omax((float)(1.0 / 32768.0), 1.0f - AMax3F1(c[0], c[1], c[2])));
  c[0] *= r; c[1] *= r; c[2] *= r;
}
/* :1012 FsrLfgaF   c += (t*a) * min(1-c, c) */
static inline void FsrLfgaF(float c[3], const float t[3], float a) {
  for (int i = 0; i < 3; ++i) c[i] += (t[i] * a) * omin(1.0f - c[i], c[i]);
}
/* :1082-1091 FsrTepdDitF.  Constant expressions are folded in double and rounded once to float (the pin used
 * for every literal expression of the header). */
static inline float FsrTepdDitF(uint32_t px, uint32_t py, uint32_t f) {
  float x = (float)(px + f);
  float y = (float)py;
  float a = (float)((1.0 + (double)sqrtf(5.0f)) / 2.0); /* GLSL sqrt(5.0) is a binary32 sqrt */
  float b = (float)(1.0 / 3.69);
  x = x * a + (y * b);
  return x - floorf(x); /* fract */
}
/* :1097-1110 FsrTepdC8F / :1113-1120 FsrTepdC10F, steps = 255 or 1023 */
static inline void FsrTepdCF(float c[3], float dit, double steps) {
  const float k = (float)steps, rk = (float)(1.0 / steps);
  for (int i = 0; i < 3; ++i) {
    float n = sqrtf(c[i]);
    n = floorf(n * k) * rk;
    float a = n * n;
    float b = n + rk;
    b = b * b;
    float r = (c[i] - b) * APrxMedRcpF1(a - b);
    c[i] = ASatF1(n + AGtZeroF1(dit - r) * rk);
  }
}

/* Stage chain on rows [y0,y1); same contract as ref_color_f in oracle/ref_wrap.cpp. */
void oracle_color_f(const float* in, int W, int H, float* out, int stages, float amount, float bias, uint32_t frame,
                    const float* noise, int nW, int nH, int nS, int nox, int noy, int y0, int y1) {
  (void)H;
#pragma omp parallel for schedule(dynamic, 4)
  for (int y = y0; y < y1; ++y)
    for (int x = 0; x < W; ++x) {
      const float* q = in + ((size_t)y * W + x) * 4;
      float c[3] = {q[0], q[1], q[2]};
      float n[4] = {0.0f, 0.0f, 0.0f, 0.0f};
      if (noise) {
        const int ny = (int)((((long long)y + noy) % nH + nH) % nH), nx = (int)((((long long)x + nox) % nW + nW) % nW);
        const float* t = noise + (((size_t)(frame % (uint32_t)nS) * nH + ny) * nW + nx) * 4;
        n[0] = t[0]; n[1] = t[1]; n[2] = t[2]; n[3] = t[3];
      }
      if (stages & ORACLE_COLOR_SRTM) FsrSrtmF(c);
      if (stages & ORACLE_COLOR_LFGA) {
        const float t[3] = {n[0] + bias, n[1] + bias, n[2] + bias};
        FsrLfgaF(c, t, amount);
      }
      if (stages & ORACLE_COLOR_SRTM_INV) FsrSrtmInvF(c);
      if (stages & (ORACLE_COLOR_TEPD_C8 | ORACLE_COLOR_TEPD_C10)) {
        const float dit = (stages & ORACLE_COLOR_DITHER_FROM_NOISE) ? ASatF1(n[3]) : FsrTepdDitF((uint32_t)x, (uint32_t)y, frame);
        FsrTepdCF(c, dit, (stages & ORACLE_COLOR_TEPD_C8) ? 255.0 : 1023.0);
      }
      float* o = out + ((size_t)y * W + x) * 4;
      o[0] = c[0]; o[1] = c[1]; o[2] = c[2]; o[3] = q[3];
    }
}

float oracle_tepd_dit_f(uint32_t x, uint32_t y, uint32_t f) { return FsrTepdDitF(x, y, f); }

/* ------------------------------------------------------------------------------------------ */
/* colour stages, half-precision entry points (ffx_fsr1.h:1017-1024, :1048-1056, :1124-1198)    */
/* every operation rounded once to binary16, like the other H restatements above               */
/* ------------------------------------------------------------------------------------------ */
static inline hf AGtZeroH1(hf m) { return hsat(hmul(m, INFINITY)); }                      /* ffx_a.h:1524 */
static inline hf AMax3H1(hf a, hf b, hf c) { return hmax(a, hmax(b, c)); }

/* :1049 FsrSrtmH / :1050 FsrSrtmInvH (ARcpH1 = correctly rounded 1/x) */
static inline void FsrSrtmH(hf c[3]) {
  hf r = hdiv(1.0f, hadd(AMax3H1(c[0], c[1], c[2]), 1.0f));
  c[0] = hmul(c[0], r); c[1] = hmul(c[1], r); c[2] = hmul(c[2], r);
}
static inline void FsrSrtmInvH(hf c[3]) {
  hf r = hdiv(1.0f, hmax(hround(1.0 / 32768.0), hsub(1.0f, AMax3H1(c[0], c[1], c[2]))));
  c[0] = hmul(c[0], r); c[1] = hmul(c[1], r); c[2] = hmul(c[2], r);
}
/* :1019 FsrLfgaH */
static inline void FsrLfgaH(hf c[3], const hf t[3], hf a) {
  for (int i = 0; i < 3; ++i) c[i] = hadd(c[i], hmul(hmul(t[i], a), hmin(hsub(1.0f, c[i]), c[i])));
}
/* :1125-1131 FsrTepdDitH: binary32 arithmetic, the result narrowed once */
static inline hf FsrTepdDitH(uint32_t px, uint32_t py, uint32_t f) { return hround((double)FsrTepdDitF(px, py, f)); }
/* :1134-1150 FsrTepdC8H / FsrTepdC10H */
static inline void FsrTepdCH(hf c[3], hf dit, double steps) {
  const hf k = hround(steps), rk = hround(1.0 / steps);
  for (int i = 0; i < 3; ++i) {
    hf n = hround(sqrt((double)c[i]));
    n = hmul(floorf(hmul(n, k)), rk);
    hf a = hmul(n, n);
    hf b = hadd(n, rk);
    b = hmul(b, b);
    hf r = hmul(hsub(c[i], b), APrxMedRcpH1(hsub(a, b)));
    c[i] = hsat(hadd(n, hmul(AGtZeroH1(hsub(dit, r)), rk)));
  }
}

void oracle_color_h(const float* in, int W, int H, float* out, int stages, float amount, float bias, uint32_t frame,
                    const float* noise, int nW, int nH, int nS, int nox, int noy, int y0, int y1) {
  (void)H;
#pragma omp parallel for schedule(dynamic, 4)
  for (int y = y0; y < y1; ++y)
    for (int x = 0; x < W; ++x) {
      const float* q = in + ((size_t)y * W + x) * 4;
      hf c[3] = {hround(q[0]), hround(q[1]), hround(q[2])};
      float n[4] = {0.0f, 0.0f, 0.0f, 0.0f};
      if (noise) {
        const int ny = (int)((((long long)y + noy) % nH + nH) % nH), nx = (int)((((long long)x + nox) % nW + nW) % nW);
        const float* t = noise + (((size_t)(frame % (uint32_t)nS) * nH + ny) * nW + nx) * 4;
        n[0] = t[0]; n[1] = t[1]; n[2] = t[2]; n[3] = t[3];
      }
      const hf hb = hround(bias), ha = hround(amount);
      if (stages & ORACLE_COLOR_SRTM) FsrSrtmH(c);
      if (stages & ORACLE_COLOR_LFGA) {
        const hf t[3] = {hadd(hround(n[0]), hb), hadd(hround(n[1]), hb), hadd(hround(n[2]), hb)};
        FsrLfgaH(c, t, ha);
      }
      if (stages & ORACLE_COLOR_SRTM_INV) FsrSrtmInvH(c);
      if (stages & (ORACLE_COLOR_TEPD_C8 | ORACLE_COLOR_TEPD_C10)) {
        const hf dit = (stages & ORACLE_COLOR_DITHER_FROM_NOISE) ? hsat(hround(n[3])) : FsrTepdDitH((uint32_t)x, (uint32_t)y, frame);
        FsrTepdCH(c, dit, (stages & ORACLE_COLOR_TEPD_C8) ? 255.0 : 1023.0);
      }
      float* o = out + ((size_t)y * W + x) * 4;
      o[0] = c[0]; o[1] = c[1]; o[2] = c[2]; o[3] = q[3];
    }
}
