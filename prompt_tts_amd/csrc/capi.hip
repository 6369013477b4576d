// ABI bookkeeping for the C library.
#include "common.h"

extern "C" int pt_abi_version(void) { return 24; }

thread_local int pt_g_last_hip_error = 0;      // per calling thread: concurrent callers do not overwrite each other's error
extern "C" const char* pt_last_hip_error(void) { return hipGetErrorString((hipError_t)pt_g_last_hip_error); }

extern "C" const char* pt_status_string(int status) {
  switch (status) {
    case PT_OK: return "ok";
    case PT_ERR_SHAPE: return "shape not supported by the kernel";
    case PT_ERR_DTYPE: return "dtype must be PT_F32 or PT_BF16";
    case PT_ERR_LAUNCH: return "HIP launch failed";
    case PT_ERR_ALIGN: return "pointer or leading dimension not 16-byte aligned";
    case PT_ERR_ARG: return "invalid argument";
    default: return "unknown status";
  }
}

extern "C" int pt_struct_size(int which) {
  switch (which) {
    case 0: return (int)sizeof(pt_operand);
    case 1: return (int)sizeof(pt_gemm_desc);
    case 2: return (int)sizeof(pt_attn_desc);
    case 3: return (int)sizeof(pt_param_seg);
    case 4: return (int)sizeof(pt_rowconv_desc);
    case 5: return (int)sizeof(pt_lstm2_desc);
    case 6: return (int)sizeof(pt_fold_seg);
    case 7: return (int)sizeof(pt_encodec_tail_desc);
    case 8: return (int)sizeof(pt_encodec_stage_desc);
    case 9: return (int)sizeof(pt_transpose_seg);
    case 10: return (int)sizeof(pt_decode_linear_desc);
    default: return -1;
  }
}
