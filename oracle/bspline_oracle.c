/*
 * TEST INFRASTRUCTURE - NOT PRODUCT CODE.
 *
 * CPU oracle for the batched B-spline evaluation path: a plain-C restatement of
 * the reference's algorithm (bspy/_spline_evaluation.py:4-27, :109-164, :205-213
 * in /root/reference), pinned against the reference's own golden tables and
 * against outputs of the reference itself (tests/golden/, tests/test_oracle.py).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library.  The product (bspy_amd/) never links, imports or calls it.
 */
#include <math.h>
#include <stdlib.h>

#define ORC_MAX_NIND 8
#define ORC_MAX_ORDER 32
#define ORC_MAX_NDEP 64

#define REAL double
#define SUF f64
#include "bspline_oracle_impl.h"
#undef REAL
#undef SUF

#define REAL float
#define SUF f32
#include "bspline_oracle_impl.h"
#undef REAL
#undef SUF

int orc_limits(int which) { return which == 0 ? ORC_MAX_NIND : which == 1 ? ORC_MAX_ORDER : ORC_MAX_NDEP; }
