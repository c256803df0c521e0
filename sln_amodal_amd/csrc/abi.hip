#include "common.h"

extern "C" int sln_abi_version(void) { return 12; }  // 12: sln_rle_to_strings (batched string codec of the evaluation hand-off); 11: loader front end (sln_label_zoom_u64, sln_label_num_objects_ragged_u64), single-part fp16 operands (parts = 1), MFMA grouped 3x3, sln_conv_wgrad_last_kernel, sln_scale_update_headroom_f32; 10: sln_conv_fwd_last_kernel (profiling label of the forward kernel a call launched; conv_fwd128x256h_kernel); 9: deferred / batched wgrad reduce, pyramid gather backward; 8: parts-only residual / mask operands, optimiser skip counter; 3: scaled split-fp16 operands (scale / amax arguments); 4: tail, optimiser, stem, crop accumulate; 5: top-k workspace; 6: scale history; 7: wgrad gw_layout

extern "C" const char *sln_error_string(int code) {
    switch (code) {
        case SLN_OK: return "ok";
        case SLN_ERR_INVALID_ARG: return "invalid argument";
        case SLN_ERR_WORKSPACE: return "workspace too small";
        case SLN_ERR_LAUNCH: return "kernel launch failed";
        case SLN_ERR_UNSUPPORTED: return "unsupported configuration";
        default: return "unknown error";
    }
}
