/*
 * TEST INFRASTRUCTURE ONLY (see msda3d_oracle_impl.h).  Builds the float and
 * double instantiations of the scalar CPU restatement into one shared object:
 *
 *   msda3d_oracle_forward_f32 / _f64, msda3d_oracle_backward_f32 / _f64
 *
 * Build:  make -C oracle      (gcc -O2 -shared -fPIC; no -ffast-math, so the
 * float build keeps IEEE evaluation order)
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#define REAL float
#define SUFFIX f32
#define FLOOR floorf
#include "msda3d_oracle_impl.h"
#undef REAL
#undef SUFFIX
#undef FLOOR

#define REAL double
#define SUFFIX f64
#define FLOOR floor
#include "msda3d_oracle_impl.h"
#undef REAL
#undef SUFFIX
#undef FLOOR
