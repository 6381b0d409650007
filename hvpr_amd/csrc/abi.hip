// ABI bookkeeping for libhvpr_amd.so (include/hvpr_amd.h).
#include "common.h"

// 2: packed memory bank (bank_packed argument); 3: bank_packed is required and holds IEEE fp16 tiles (v_mfma_f32_16x16x32_f16) +
// channel maxima; 4: the packed bank ends in 128 floats (64 channel maxima, their sum and maximum: hvpr_memory_bank_packed_floats grew
// by 64) and the voxelizer workspace holds the one-launch index kernel's per-cell words and barrier flags (hvpr_voxelize_workspace_bytes
// grew) — a caller that sized either buffer with a version-3 formula is too small: always size them with the two functions; 5: adds
// hvpr_bn_train_affine_f32 and hvpr_bn_relu_fwd / bwd_slice_nhwc_f32; hvpr_conv2d_wino_wgrad_nhwc_f32 answers HVPR_ERR_UNSUPPORTED for images of 2 GB and more (32-bit offsets
// inside an image; its workspace size is unchanged); 6: hvpr_encode_fwd_f32 takes index_mode (the one-launch index kernel is opt-in and guarded),
// hvpr_voxelize_workspace_status, HVPR_ERR_TIMEOUT
extern "C" int hvpr_abi_version(void) { return 6; }

extern "C" const char *hvpr_status_string(int status) {
    switch (status) {
        case HVPR_OK: return "ok";
        case HVPR_ERR_INVALID_ARG: return "invalid argument";
        case HVPR_ERR_UNSUPPORTED: return "unsupported shape for this build of the kernels";
        case HVPR_ERR_WORKSPACE: return "workspace too small";
        case HVPR_ERR_LAUNCH: return "HIP launch failed";
        case HVPR_ERR_TIMEOUT: return "a one-launch index kernel gave up a wait: reset the voxelizer workspace";
        default: return "unknown status";
    }
}

// SyncBatchNorm hook (include/hvpr_amd.h): the only state this library keeps between calls
namespace { hvpr_allreduce_fn g_bn_allreduce = nullptr; void *g_bn_allreduce_ctx = nullptr; }

extern "C" void hvpr_set_batchnorm_allreduce(hvpr_allreduce_fn fn, void *ctx) { g_bn_allreduce = fn; g_bn_allreduce_ctx = ctx; }

int hvpr_i_bn_allreduce(double *buf, int n, hipStream_t s) {
    if (!g_bn_allreduce) return 0;
    return g_bn_allreduce(buf, n, (hvpr_stream_t)s, g_bn_allreduce_ctx);
}
