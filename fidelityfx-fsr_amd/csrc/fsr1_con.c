/*
 * Host-side constant setup of the FSR 1.0 hot path: FsrEasuCon, FsrEasuConOffset, FsrRcasCon.
 *
 * Bit-exact replacement of the A_CPU build of the reference (ffx-fsr/ffx_fsr1.h:156-225,
 * :662-672 with the ffx_a.h helpers AU1_AF1 :141, ARcpF1 :326, AExp2F1 :283, AU1_AH1_AF1
 * :482-549, AU1_AH2_AF2 :552).  Must be compiled without FMA contraction and without
 * fast-math (the Makefile passes -ffp-contract=off): "0.5*vp*rcp(out)-0.5" is two roundings in
 * the reference and has to stay two roundings here.
 */
#include <math.h>
#include <string.h>

#include "fsr1_hip.h"

static uint32_t bits_of(float f) {
  uint32_t u;
  memcpy(&u, &f, sizeof u);
  return u;
}

/* The reference multiplies by a reciprocal instead of dividing (ffx_fsr1.h:171-174), so
 * 2560/3840 and 1440/2160 differ in the last bit; keep that. */
static float rcp(float x) { return 1.0f / x; }

void FsrEasuCon(uint32_t* con0, uint32_t* con1, uint32_t* con2, uint32_t* con3, float inputViewportInPixelsX,
                float inputViewportInPixelsY, float inputSizeInPixelsX, float inputSizeInPixelsY,
                float outputSizeInPixelsX, float outputSizeInPixelsY) {
  const float rOutX = rcp(outputSizeInPixelsX), rOutY = rcp(outputSizeInPixelsY);
  const float rInX = rcp(inputSizeInPixelsX), rInY = rcp(inputSizeInPixelsY);
  /* con0: output pixel -> input pixel position (scale, then half-texel bias) */
  con0[0] = bits_of(inputViewportInPixelsX * rOutX);
  con0[1] = bits_of(inputViewportInPixelsY * rOutY);
  con0[2] = bits_of(0.5f * inputViewportInPixelsX * rOutX - 0.5f);
  con0[3] = bits_of(0.5f * inputViewportInPixelsY * rOutY - 0.5f);
  /* con1: texel size, and the centre of gather 0 relative to the top-left of texel 'f' */
  con1[0] = bits_of(rInX);
  con1[1] = bits_of(rInY);
  con1[2] = bits_of(1.0f * rInX);
  con1[3] = bits_of(-1.0f * rInY);
  /* con2/con3: centres of gathers 1,2,3 relative to gather 0 */
  con2[0] = bits_of(-1.0f * rInX);
  con2[1] = bits_of(2.0f * rInY);
  con2[2] = bits_of(1.0f * rInX);
  con2[3] = bits_of(2.0f * rInY);
  con3[0] = bits_of(0.0f * rInX);
  con3[1] = bits_of(4.0f * rInY);
  con3[2] = 0;
  con3[3] = 0;
}

void FsrEasuConOffset(uint32_t* con0, uint32_t* con1, uint32_t* con2, uint32_t* con3, float inputViewportInPixelsX,
                      float inputViewportInPixelsY, float inputSizeInPixelsX, float inputSizeInPixelsY,
                      float outputSizeInPixelsX, float outputSizeInPixelsY, float inputOffsetInPixelsX,
                      float inputOffsetInPixelsY) {
  FsrEasuCon(con0, con1, con2, con3, inputViewportInPixelsX, inputViewportInPixelsY, inputSizeInPixelsX,
             inputSizeInPixelsY, outputSizeInPixelsX, outputSizeInPixelsY);
  con0[2] = bits_of(0.5f * inputViewportInPixelsX * rcp(outputSizeInPixelsX) - 0.5f + inputOffsetInPixelsX);
  con0[3] = bits_of(0.5f * inputViewportInPixelsY * rcp(outputSizeInPixelsY) - 0.5f + inputOffsetInPixelsY);
}

/* float -> binary16 bit pattern by truncation of the mantissa (no rounding), denormals produced,
 * everything at or above 65520-ish, infinities and NaNs clamped to +-65504.  The reference does it
 * with two 512-entry tables indexed by sign|exponent; the tables encode exactly these four ranges. */
uint32_t AU1_AH1_AF1(float f) {
  const uint32_t u = bits_of(f);
  const uint32_t sign = (u >> 16) & 0x8000u;
  const int32_t exp = (int32_t)((u >> 23) & 0xffu) - 127; /* unbiased */
  const uint32_t man = u & 0x007fffffu;
  if (exp < -24) return sign;                       /* underflows even the smallest denormal */
  if (exp < -14) {                                  /* half denormal: hidden bit becomes explicit */
    const int sh = -14 - exp;                       /* 1..10 */
    return sign + (0x0400u >> sh) + (man >> (13 + sh));
  }
  if (exp <= 15) return sign + ((uint32_t)(exp + 15) << 10) + (man >> 13);
  return sign + 0x7bffu;
}

void FsrRcasCon(uint32_t* con, float sharpness) {
  const float linear = exp2f(-sharpness); /* stops -> linear */
  const uint32_t h = AU1_AH1_AF1(linear);
  con[0] = bits_of(linear);
  con[1] = h + (h << 16);
  con[2] = 0;
  con[3] = 0;
}
