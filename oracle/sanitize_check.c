/* TEST INFRASTRUCTURE — sanitizer job (the reference has none; this is ours).
 * Built with -fsanitize=address,undefined together with oracle/fsr1_oracle.c and the product's host constant code
 * (fidelityfx-fsr_amd/csrc/fsr1_con.c), it runs every CPU entry point the tests rely on over ragged, 1x1 and
 * ratio-extreme shapes (where clamp-to-edge and the out-of-bounds-is-zero rule do all the work) and cross-checks the two
 * constant-setup implementations bit for bit.  Any out-of-bounds access, signed overflow or misaligned access aborts. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "fsr1_hip.h"

void oracle_FsrEasuCon(uint32_t* con16, float vpX, float vpY, float inX, float inY, float outX, float outY);
void oracle_FsrEasuConOffset(uint32_t* con16, float vpX, float vpY, float inX, float inY, float outX, float outY, float offX, float offY);
void oracle_FsrRcasCon(uint32_t* con, float sharpness);
uint32_t oracle_AU1_AH1_AF1(float f);
void oracle_easu_f(const float* in, int inW, int inH, float* out, int outW, int outH, const uint32_t* con16, int flags, int y0, int y1);
void oracle_easu_h(const float* in, int inW, int inH, float* out, int outW, int outH, const uint32_t* con16, int flags, int y0, int y1);
void oracle_rcas_f(const float* in, int W, int H, float* out, const uint32_t* con, int flags, int y0, int y1);
void oracle_rcas_h(const float* in, int W, int H, float* out, const uint32_t* con, int flags, int y0, int y1);
void oracle_color_f(const float* in, int W, int H, float* out, int stages, float amount, float bias, uint32_t frame, const float* noise, int nW,
                    int nH, int nS, int offX, int offY, int y0, int y1);

static float* frame(int w, int h, uint32_t seed) {
  float* p = (float*)malloc((size_t)w * h * 4 * sizeof(float));
  for (size_t i = 0; i < (size_t)w * h * 4; ++i) {
    seed = seed * 1664525u + 1013904223u;
    p[i] = (i & 3) == 3 ? 1.0f : (float)(seed >> 8) * (1.0f / 16777216.0f);
  }
  return p;
}

int main(void) {
  static const int shapes[][4] = {{1, 1, 2, 2}, {1, 1, 7, 5}, {5, 3, 17, 9}, {33, 9, 66, 18}, {37, 23, 48, 29}, {64, 16, 21, 5}, {3, 40, 13, 41}};
  int fails = 0;
  for (size_t s = 0; s < sizeof shapes / sizeof shapes[0]; ++s) {
    const int iw = shapes[s][0], ih = shapes[s][1], ow = shapes[s][2], oh = shapes[s][3];
    uint32_t con[16], con2[16], rc[4], rc2[4];
    oracle_FsrEasuCon(con, (float)iw, (float)ih, (float)iw, (float)ih, (float)ow, (float)oh);
    FsrEasuCon(con2, con2 + 4, con2 + 8, con2 + 12, (float)iw, (float)ih, (float)iw, (float)ih, (float)ow, (float)oh);
    fails += memcmp(con, con2, sizeof con) != 0;
    oracle_FsrEasuConOffset(con, (float)iw * 0.5f, (float)ih * 0.5f, (float)iw, (float)ih, (float)ow, (float)oh, 1.0f, 0.0f);
    FsrEasuConOffset(con2, con2 + 4, con2 + 8, con2 + 12, (float)iw * 0.5f, (float)ih * 0.5f, (float)iw, (float)ih, (float)ow, (float)oh, 1.0f, 0.0f);
    fails += memcmp(con, con2, sizeof con) != 0;
    oracle_FsrEasuCon(con, (float)iw, (float)ih, (float)iw, (float)ih, (float)ow, (float)oh);
    for (int k = 0; k < 3; ++k) {
      const float stops = (float)k * 0.75f;
      oracle_FsrRcasCon(rc, stops);
      FsrRcasCon(rc2, stops);
      fails += memcmp(rc, rc2, sizeof rc) != 0;
      fails += oracle_AU1_AH1_AF1(stops + 0.1f) != AU1_AH1_AF1(stops + 0.1f);
    }
    float* in = frame(iw, ih, 0x1234u + (uint32_t)s);
    float* mid = (float*)malloc((size_t)ow * oh * 4 * sizeof(float));
    float* out = (float*)malloc((size_t)ow * oh * 4 * sizeof(float));
    for (int flags = 0; flags < 8; flags += 4) {  /* bit 2 = HDR square */
      oracle_easu_f(in, iw, ih, mid, ow, oh, con, flags, 0, oh);
      oracle_easu_h(in, iw, ih, out, ow, oh, con, flags, 0, oh);
    }
    for (int flags = 0; flags < 8; ++flags) {       /* denoise, alpha pass-through, HDR square */
      oracle_rcas_f(mid, ow, oh, out, rc, flags, 0, oh);
      oracle_rcas_h(mid, ow, oh, out, rc, flags, 0, oh);
    }
    float* noise = frame(8, 8, 99u);
    oracle_color_f(mid, ow, oh, out, 1 | 2 | 4, 0.25f, -0.5f, 3u, noise, 8, 8, 1, 5, 7, 0, oh);
    oracle_color_f(mid, ow, oh, out, 2 | 8 | 32, 0.25f, -0.5f, 3u, noise, 8, 8, 1, -3, 9, 0, oh);
    for (size_t i = 0; i < (size_t)ow * oh * 4; ++i) fails += out[i] != out[i];  /* no NaN out of bounded input */
    free(noise); free(in); free(mid); free(out);
  }
  printf("sanitize_check: %d failure(s)\n", fails);
  return fails != 0;
}
