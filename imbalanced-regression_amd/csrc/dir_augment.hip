// Training-time image augmentation of imdb-wiki-dir/datasets.py:38-53 on the GPU: what torchvision's
//   RandomCrop(S, padding=16) -> RandomHorizontalFlip() -> ToTensor() -> Normalize([.5]*3, [.5]*3)
// does to a decoded, already resized uint8 image, for a whole batch in one launch, with the stem's input layout and dtype
// as the output (NHWC = a channels_last [B, 3, S, S] tensor, float32 or bf16: the bf16 form is what the MFMA stem kernel
// reads, so the separate cast pass of the host path disappears). The random draws (crop offset, flip) are INPUTS: the host
// draws them with the reference's generator calls (dirhip/datasets.py), so the augmentation is reproducible and testable.
//   out[b, y, x, c] = ((float)src / 255 - 0.5) / 0.5,   src = img[b, top + y - pad, left + xs - pad, c] or 0 outside,
//   xs = flip ? S - 1 - x : x                            (pad -> crop -> flip, in that order)
// float32 arithmetic exactly as ToTensor / Normalize execute it (u8 -> float, / 255, - mean, / std; -ffp-contract=off).
// HBM-trivial: 3 B read, 6-12 B written per pixel; one thread per output pixel.
#include "dir_common.h"

namespace {
__device__ __forceinline__ uint32_t au_f2bf(float f) {
    const uint32_t u = __float_as_uint(f);
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;                // values are in [-1, 1]: no NaN / Inf handling needed
}

template <bool BF16>
__global__ void __launch_bounds__(DIR_TPB)
augment_kernel(const uint8_t* __restrict__ img, const int* __restrict__ params, void* __restrict__ out, int B, int S, int pad) {
    const long long total = (long long)B * S * S;
    for (long long i = (long long)blockIdx.x * DIR_TPB + threadIdx.x; i < total; i += (long long)gridDim.x * DIR_TPB) {
        const int x = (int)(i % S);
        const long long r = i / S;
        const int y = (int)(r % S), b = (int)(r / S);
        int top = pad, left = pad, flip = 0;                      // no parameters: the evaluation transform (no crop, no flip)
        if (params) { top = params[3 * b]; left = params[3 * b + 1]; flip = params[3 * b + 2]; }
        const int sy = top + y - pad, sx = left + (flip ? S - 1 - x : x) - pad;
        uint32_t v0 = 0, v1 = 0, v2 = 0;
        if ((unsigned)sy < (unsigned)S && (unsigned)sx < (unsigned)S) {
            const uint8_t* p = img + (((size_t)b * S + sy) * S + sx) * 3;
            v0 = p[0]; v1 = p[1]; v2 = p[2];
        }
        const float f0 = ((float)v0 / 255.0f - 0.5f) / 0.5f, f1 = ((float)v1 / 255.0f - 0.5f) / 0.5f, f2 = ((float)v2 / 255.0f - 0.5f) / 0.5f;
        if (BF16) {
            uint16_t* o = static_cast<uint16_t*>(out) + i * 3;
            o[0] = (uint16_t)au_f2bf(f0); o[1] = (uint16_t)au_f2bf(f1); o[2] = (uint16_t)au_f2bf(f2);
        } else {
            float* o = static_cast<float*>(out) + i * 3;
            o[0] = f0; o[1] = f1; o[2] = f2;
        }
    }
}
}  // namespace

extern "C" int dir_augment_u8(const void* img, const int* params, void* out, int dtype, int B, int S, int pad, dir_stream_t stream) {
    DIR_RETURN_IF(!img || !out || B <= 0 || S <= 0 || pad < 0, DIR_EINVAL);
    DIR_RETURN_IF(dtype != DIR_F32 && dtype != DIR_BF16, DIR_EUNSUPPORTED);
    const long long total = (long long)B * S * S;
    long long blocks = (total + DIR_TPB - 1) / DIR_TPB;
    if (blocks > 65536) blocks = 65536;
    if (dtype == DIR_BF16)
        hipLaunchKernelGGL(augment_kernel<true>, dim3((unsigned)blocks), dim3(DIR_TPB), 0, dir_s(stream), static_cast<const uint8_t*>(img), params, out, B, S, pad);
    else
        hipLaunchKernelGGL(augment_kernel<false>, dim3((unsigned)blocks), dim3(DIR_TPB), 0, dir_s(stream), static_cast<const uint8_t*>(img), params, out, B, S, pad);
    DIR_LAUNCH_CHECK();
    return DIR_OK;
}
