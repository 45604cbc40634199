// The small-tile product kernel's 32-row tiles (gemm16_kernel.h; routing and the why in gemm16.hip): a translation unit of its own so that
// the tile heights compile in parallel.
#include "gemm16_kernel.h"

int launch16_rb2(int nt, const g16::Gemm16Args& a, int act, bool add, bool wkn, hipStream_t s) {
  switch (nt) {
    case 8: return g16::launch16_nt<2, 8>(a, act, add, wkn, s);
    case 12: return g16::launch16_nt<2, 12>(a, act, add, wkn, s);
    default: return g16::launch16_nt<2, 16>(a, act, add, wkn, s);
  }
}
