// asrfe_encoder_forward_static: the encoder forward for callers that cannot allocate device memory (code under HIP
// graph capture, tight serving loops, harnesses that forbid hipMalloc).  It is asrfe_encoder_forward with the
// workspace taken from a device array that is part of the library's code object instead of from the caller.
//
// The array exists once per GPU (every device that loads the code object gets its own copy at its own address) for
// the life of the process.  Its device address is looked up once per device, in asrfe_create with that device
// current, and kept in a per-device table; from then on host code uses that pointer only.  The array's NAME is a
// host-side stand-in, never a device address: it must not be passed to a kernel or a memset.
//
// All calls on one GPU share the array, so two calls that overlap on one GPU (different streams or threads) race;
// calls on one stream, or ordered by events, are fine.  Nothing in the workspace has to start at zero: every buffer
// the encoder carves from it is written in full by one kernel before another reads it, so the contents a previous
// (possibly larger) call left behind are never observed.
#include "common.h"

namespace asrfe {

constexpr int64_t kStaticScratchBytes = ASRFE_STATIC_SCRATCH_BYTES;
constexpr int kMaxDevices = 64;

static __device__ __align__(256) __attribute__((used)) unsigned char scratch[kStaticScratchBytes];

static void* g_scratch_ptr[kMaxDevices] = {};  // device address of `scratch` on each device; written in asrfe_create

// Called by asrfe_create with the context's device current (asrfe_create runs before any compute call on the context;
// a second context on the same device stores the same address again).
int32_t prepare_static_scratch(asrfe_ctx* ctx) {
    if (ctx->device < 0 || ctx->device >= kMaxDevices)
        return set_err(ctx, ASRFE_ERR_BAD_ARG, "asrfe_create: device ordinal beyond the static scratch table");
    void* ptr = nullptr;
    const hipError_t e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
    if (e != hipSuccess || ptr == nullptr)
        return set_err(ctx, ASRFE_ERR_HIP, std::string("hipGetSymbolAddress(scratch): ") + hipGetErrorString(e));
    if (reinterpret_cast<uintptr_t>(ptr) % 256 != 0)
        return set_err(ctx, ASRFE_ERR_HIP, "static scratch array is not 256-byte aligned");
    g_scratch_ptr[ctx->device] = ptr;
    return ASRFE_OK;
}

}  // namespace asrfe

using namespace asrfe;

extern "C" int32_t asrfe_encoder_forward_static(asrfe_ctx* ctx, const float* feats, const int32_t* lens, int32_t batch,
                                                int32_t tp, void* out, int32_t out_dtype, void* stream) {
    if (!ctx) return ASRFE_ERR_BAD_ARG;
    if (!ctx->enc_loaded) return set_err(ctx, ASRFE_ERR_NOT_LOADED, "asrfe_encoder_forward_static: encoder weights not loaded");
    if (batch < 0 || batch > 16384 || tp < 0 || tp > 8192)
        return set_err(ctx, ASRFE_ERR_SHAPE, "asrfe_encoder_forward_static: batch in 0..16384, tp in 0..8192");
    if (batch == 0 || tp == 0) return ASRFE_OK;
    const int64_t need = encoder_workspace_bytes(ctx, batch, tp);
    if (need > kStaticScratchBytes) {
        char buf[224];
        std::snprintf(buf, sizeof(buf),
                      "asrfe_encoder_forward_static: [%d x %d] needs %lld bytes of scratch, the static array holds %lld "
                      "(use asrfe_encoder_forward with a caller workspace)",
                      batch, tp, static_cast<long long>(need), static_cast<long long>(kStaticScratchBytes));
        return set_err(ctx, ASRFE_ERR_WORKSPACE, buf);
    }
    if (!feats || !lens || !out) return set_err(ctx, ASRFE_ERR_BAD_ARG, "asrfe_encoder_forward_static: null argument");
    if (out_dtype != ASRFE_DT_F32 && out_dtype != ASRFE_DT_BF16 && out_dtype != ASRFE_DT_F16)
        return set_err(ctx, ASRFE_ERR_BAD_ARG, "asrfe_encoder_forward_static: bad out_dtype");
    if (reinterpret_cast<uintptr_t>(feats) % 16 != 0 || reinterpret_cast<uintptr_t>(out) % 16 != 0)
        return set_err(ctx, ASRFE_ERR_BAD_ARG, "asrfe_encoder_forward_static: feats / out must be 16-byte aligned");
    void* ws = ctx->device >= 0 && ctx->device < kMaxDevices ? g_scratch_ptr[ctx->device] : nullptr;
    if (!ws) return set_err(ctx, ASRFE_ERR_HIP, "asrfe_encoder_forward_static: no scratch address for this device");
    return encoder_forward(ctx, feats, lens, batch, tp, out, out_dtype, ws, kStaticScratchBytes,
                           static_cast<hipStream_t>(stream));
}
