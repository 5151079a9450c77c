/* TEST INFRASTRUCTURE — not product code.
 *
 * The reference's A_CPU build of the constant setup, compiled from the headers where they lie
 * (-I/root/reference/ffx-fsr; see build_ref.sh): FsrEasuCon / FsrEasuConOffset (ffx_fsr1.h:156-225),
 * FsrRcasCon (ffx_fsr1.h:662-672) and the truncating float->half table conversion AU1_AH1_AF1
 * (ffx_a.h:482-549).  Exported under ref_* names so it can sit in one process next to the product
 * library, which exports the reference spellings. */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#define A_CPU 1
#include "ffx_a.h"
#include "ffx_fsr1.h"

void ref_FsrEasuCon(uint32_t* con16, float vpW, float vpH, float inW, float inH, float outW, float outH) {
  FsrEasuCon(con16, con16 + 4, con16 + 8, con16 + 12, vpW, vpH, inW, inH, outW, outH);
}
void ref_FsrEasuConOffset(uint32_t* con16, float vpW, float vpH, float inW, float inH, float outW, float outH,
                          float offX, float offY) {
  FsrEasuConOffset(con16, con16 + 4, con16 + 8, con16 + 12, vpW, vpH, inW, inH, outW, outH, offX, offY);
}
void ref_FsrRcasCon(uint32_t* con4, float sharpness) { FsrRcasCon(con4, sharpness); }
uint32_t ref_AU1_AH1_AF1(float f) { return AU1_AH1_AF1(f); }
