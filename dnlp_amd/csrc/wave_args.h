// Template-specialised batch solver: the argument block of a launch (wave_batch.h: the library's own kernels;
// wave_codegen.h: the kernels compiled per template at run time — which is why this text is free of standard-library
// includes and also travels inside the library as wave_args_src.inc).
#pragma once
#include "ipm_options.h"

namespace dnlp {

struct WaveArgs {
  const i32* blk = nullptr;          // the plan block (device memory)
  const int16_t* blk16 = nullptr;    // ... narrowed to 16 bits (null when an entry does not fit)
  int blk_ints = 0;
  const unsigned* gen = nullptr;     // work tables of the generated LDL^T phases (wave_gen.h; per-template kernels only)
  int gen_words = 0;
  const double* rows = nullptr;      // batch x row_doubles instance rows (batch.h slab layout)
  i64 row_doubles = 0;
  int batch = 0;
  double* state = nullptr;           // state in global memory: (grid x NW) x state_doubles
  double* park = nullptr;            // (grid x NW) x park_doubles: polish()'s parking place (wave_plan.h wave_park_doubles)
  i64 park_doubles = 0;
  i64 state_doubles = 0;
  IpmOptions opt;
  i64 fallback_max_n = 0;
  double *x_out = nullptr, *obj_out = nullptr, *multg_out = nullptr, *zl_out = nullptr, *zu_out = nullptr;
  int *status_out = nullptr, *iters_out = nullptr, *nfact_out = nullptr;
  double* times_out = nullptr;
  const double *ws_g = nullptr, *ws_l = nullptr, *ws_u = nullptr;
  int* next = nullptr;
  const int* order = nullptr;
  unsigned long long* prof = nullptr;   // kWaveProfSlots + 1 counters of a -DDNLP_WAVE_PROF build (last: iterations)
};

}  // namespace dnlp
