// mpcqp_quad4w.hip -- the instantiations for nx = 5 .. 8 of the four-rows-per-lane copy of the four-per-wavefront kernel
// (csrc/mpcqp_quad4.hip: 33 .. 64 rows at n <= 16), compiled as a unit of their own for the build time. Same source, same reference code
// replaced (qpmpc/mpc_qp.py:53-149, qpmpc/solve_mpc.py:43).
#define MPCQP_QUAD_WIDE_UNIT 1
#include "mpcqp_quad4.hip"
