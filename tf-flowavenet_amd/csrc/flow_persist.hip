// One flow of the small-M chain as one launch (round 5): kernel and launcher in flow_persist.h; a translation unit of its
// own (five instantiations of a large kernel: flow_kernels.hip already takes four minutes to compile).
#include "common.h"
#include "gemm_ring.h"
#include "fwn_internal.h"
#include "tail_zero_prob.h"
#include "flow_persist.h"
