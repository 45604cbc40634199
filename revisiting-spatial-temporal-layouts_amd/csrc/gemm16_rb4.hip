// The small-tile product kernel's 64-row tiles (gemm16_kernel.h; routing and the why in gemm16.hip): a translation unit of its own so that
// the tile heights compile in parallel.
#include "gemm16_kernel.h"

int launch16_rb4(int nt, const g16::Gemm16Args& a, int act, bool add, bool wkn, hipStream_t s) {
  switch (nt) {
    case 4: return g16::launch16_nt<4, 4>(a, act, add, wkn, s);
    case 6: return g16::launch16_nt<4, 6>(a, act, add, wkn, s);
    case 8: return g16::launch16_nt<4, 8>(a, act, add, wkn, s);
    case 10: return g16::launch16_nt<4, 10>(a, act, add, wkn, s);
    case 12: return g16::launch16_nt<4, 12>(a, act, add, wkn, s);
    default: return g16::launch16_nt<4, 16>(a, act, add, wkn, s);
  }
}
