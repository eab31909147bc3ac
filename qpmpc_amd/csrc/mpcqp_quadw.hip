// mpcqp_quadw.hip -- the four-problems-per-wavefront kernel's instantiations for nx = 5 .. 16 (csrc/mpcqp_quad.hip: the general build
// with three / four operand registers per step for nx = 5, 6, with the operands streamed per step for nx = 7 .. 16 in the padded sizes
// 8, 12, 16), compiled as a unit of their own so that the library builds in two minutes per unit instead of four for one. Same source,
// same reference code replaced (qpmpc/mpc_qp.py:53-149, qpmpc/solve_mpc.py:43).
#define MPCQP_QUAD_WIDE_UNIT 1
#include "mpcqp_quad.hip"
