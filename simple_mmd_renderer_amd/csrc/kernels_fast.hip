// kernels_fast.hip -- the deform / frame / shared-morph kernels of kernels.hip once more, with multiply-add contraction allowed
// (models created with MMDX_CREATE_FAST_MATH): launch_deform_fast, launch_frame_fast, launch_morph_apply_fast, prepare_kernels_fast.
// The library's default kernels stay uncontracted and bit-identical to the reference.
#define MMDX_FAST_MATH 1
#include "kernels.hip"
