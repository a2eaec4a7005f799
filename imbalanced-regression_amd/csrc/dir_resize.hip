// §8f-4  transforms.Resize((S, S)) of the reference's transform chain (imdb-wiki-dir/datasets.py:41,49) on the GPU, for a RAGGED
// batch of decoded uint8 RGB images: loader workers only decode; the images travel over PCIe at their file size and are resized
// here — bit for bit what Pillow's bilinear resize returns (torchvision's Resize on a PIL image = Image.resize(BILINEAR); algorithm
// = Pillow src/libImaging/Resample.c, restated in oracle/resize_oracle.py and pinned on Pillow itself):
//   * per axis and output index: support window + triangle-filter weights in FLOAT64, antialiased when shrinking
//     (support = max(in / out, 1)), normalised by their sequential sum, then fixed point with 22 fractional bits;
//   * horizontal pass -> uint8 intermediate (rounded, clipped) -> vertical pass; int32 accumulators starting at 1 << 21;
//     a pass whose size does not change is a copy.
// Integer / byte work, HBM-trivial next to the training step (79 MB per 256 images of 320 x 320): one thread per output
// pixel (3 channels), the taps of neighbouring threads overlap in L1 / L2. Three launches: coefficient tables (float64,
// -ffp-contract=off: one rounding per operation, as the C code), horizontal, vertical.
//   table [B][4] int64 = (byte offset of the image in src, H, W, byte offset of its [H][S][3] intermediate in the workspace's tmp area)
#include "dir_common.h"

namespace {
constexpr int RZ_PRECISION_BITS = 32 - 8 - 2;

__device__ __forceinline__ double rz_tri(double x) {
    if (x < 0.0) x = -x;
    return x < 1.0 ? 1.0 - x : 0.0;
}

// grid (B, 2): axis 0 = horizontal (in = W), 1 = vertical (in = H). bounds [B][2][S][2], kk [B][2][S][kmax]
__global__ void __launch_bounds__(DIR_TPB)
resize_coeffs_kernel(const long long* __restrict__ table, int S, int kmax, int* __restrict__ bounds, int* __restrict__ kk) {
    const int b = blockIdx.x, axis = blockIdx.y;
    const int in_size = (int)table[4 * b + (axis == 0 ? 2 : 1)];
    const double scale = (double)in_size / (double)S;
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = 1.0 * filterscale;
    const double ss = 1.0 / filterscale;
    int* bo = bounds + ((size_t)(b * 2 + axis) * S) * 2;
    int* ko = kk + ((size_t)(b * 2 + axis) * S) * kmax;
    for (int xx = threadIdx.x; xx < S; xx += DIR_TPB) {
        const double center = 0.0 + ((double)xx + 0.5) * scale;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        if (xmax > kmax) xmax = kmax;                               // (cannot happen when the host passed the batch's kmax)
        double ww = 0.0;
        for (int x = 0; x < xmax; ++x) ww += rz_tri(((double)(x + xmin) - center + 0.5) * ss);
        for (int x = 0; x < kmax; ++x) {
            double w = x < xmax ? rz_tri(((double)(x + xmin) - center + 0.5) * ss) : 0.0;
            if (x < xmax && ww != 0.0) w /= ww;
            ko[(size_t)xx * kmax + x] = w < 0.0 ? (int)(-0.5 + w * (double)(1 << RZ_PRECISION_BITS)) : (int)(0.5 + w * (double)(1 << RZ_PRECISION_BITS));
        }
        bo[2 * xx] = xmin; bo[2 * xx + 1] = xmax;
    }
}

__device__ __forceinline__ uint8_t rz_clip8(int v) {
    v >>= RZ_PRECISION_BITS;
    return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// The table lives in device memory, so the host entry point cannot validate it without a sync: each kernel checks the entry it works on
// against what the launch was sized for (hmax rows of the grid, kmax taps per window, the source buffer's src_bytes, the tmp area behind the
// coefficient tables) and leaves an image it does not fit untouched instead of reading or writing out of bounds (dirhip.datasets.DeviceResize validates the same on the host before the launch).
__device__ __forceinline__ bool rz_entry_fits(int H, int W, long long soff, long long toff, int S, int hmax, int kmax, long long src_bytes, long long tmp_bytes) {
    // reads: the image inside src, windows of at most kmax taps (the coefficient tables' row length) on both axes; writes: the tmp area
    const long long kw = W <= S ? 3 : 2 * ((W + S - 1) / S) + 1, kh = H <= S ? 3 : 2 * ((H + S - 1) / S) + 1;      // dir_resize_ksize
    return H > 0 && W > 0 && H <= hmax && kw <= kmax && kh <= kmax && soff >= 0 && soff + (long long)H * W * 3 <= src_bytes &&
           toff >= 0 && toff + (long long)H * S * 3 <= tmp_bytes;
}

// horizontal: tmp[b][y][xx][c] = clip8((1 << 21) + sum_x src[b][y][xmin + x][c] * k[xx][x]); grid (ceil(Hmax * S / 256), B)
__global__ void __launch_bounds__(DIR_TPB)
resize_h_kernel(const uint8_t* __restrict__ src, const long long* __restrict__ table, int S, int hmax, int kmax, long long src_bytes, long long tmp_bytes,
                const int* __restrict__ bounds, const int* __restrict__ kk, uint8_t* __restrict__ tmp) {
    const int b = blockIdx.y;
    const long long soff = table[4 * b], toff = table[4 * b + 3];
    const int H = (int)table[4 * b + 1], W = (int)table[4 * b + 2];
    if (!rz_entry_fits(H, W, soff, toff, S, hmax, kmax, src_bytes, tmp_bytes)) return;     // a table entry the launch was not sized for: nothing is read or written for that image
    const long long i = (long long)blockIdx.x * DIR_TPB + threadIdx.x;
    if (i >= (long long)H * S) return;
    const int y = (int)(i / S), xx = (int)(i - (long long)y * S);
    const uint8_t* row = src + soff + (size_t)y * W * 3;
    uint8_t* o = tmp + toff + ((size_t)y * S + xx) * 3;
    if (W == S) { o[0] = row[3 * xx]; o[1] = row[3 * xx + 1]; o[2] = row[3 * xx + 2]; return; }     // pass skipped: no rounding
    const int* bo = bounds + ((size_t)(b * 2) * S + xx) * 2;
    const int* k = kk + ((size_t)(b * 2) * S + xx) * kmax;
    const int xmin = bo[0], n = bo[1];
    int a0 = 1 << (RZ_PRECISION_BITS - 1), a1 = a0, a2 = a0;
    const uint8_t* p = row + (size_t)xmin * 3;
    for (int x = 0; x < n; ++x) {
        const int kv = k[x];
        a0 += (int)p[3 * x] * kv; a1 += (int)p[3 * x + 1] * kv; a2 += (int)p[3 * x + 2] * kv;
    }
    o[0] = rz_clip8(a0); o[1] = rz_clip8(a1); o[2] = rz_clip8(a2);
}

// vertical: out[b][yy][xx][c] = clip8((1 << 21) + sum_y tmp[b][ymin + y][xx][c] * k[yy][y]); grid (ceil(S * S / 256), B)
__global__ void __launch_bounds__(DIR_TPB)
resize_v_kernel(const uint8_t* __restrict__ tmp, const long long* __restrict__ table, int S, int hmax, int kmax, long long src_bytes, long long tmp_bytes,
                const int* __restrict__ bounds, const int* __restrict__ kk, uint8_t* __restrict__ out) {
    const int b = blockIdx.y;
    const long long toff = table[4 * b + 3];
    const int H = (int)table[4 * b + 1];
    if (!rz_entry_fits(H, (int)table[4 * b + 2], table[4 * b], toff, S, hmax, kmax, src_bytes, tmp_bytes)) return;
    const int i = blockIdx.x * DIR_TPB + threadIdx.x;
    if (i >= S * S) return;
    const int yy = i / S, xx = i - yy * S;
    uint8_t* o = out + (((size_t)b * S + yy) * S + xx) * 3;
    const uint8_t* col = tmp + toff + (size_t)xx * 3;
    const size_t pitch = (size_t)S * 3;
    if (H == S) { const uint8_t* p = col + (size_t)yy * pitch; o[0] = p[0]; o[1] = p[1]; o[2] = p[2]; return; }
    const int* bo = bounds + ((size_t)(b * 2 + 1) * S + yy) * 2;
    const int* k = kk + ((size_t)(b * 2 + 1) * S + yy) * kmax;
    const int ymin = bo[0], n = bo[1];
    int a0 = 1 << (RZ_PRECISION_BITS - 1), a1 = a0, a2 = a0;
    const uint8_t* p = col + (size_t)ymin * pitch;
    for (int y = 0; y < n; ++y) {
        const int kv = k[y];
        a0 += (int)p[0] * kv; a1 += (int)p[1] * kv; a2 += (int)p[2] * kv;
        p += pitch;
    }
    o[0] = rz_clip8(a0); o[1] = rz_clip8(a1); o[2] = rz_clip8(a2);
}

size_t rz_coef_bytes(int B, int S, int kmax) { return dir_align_up(sizeof(int) * (size_t)B * 2 * S * (2 + (size_t)kmax), 256); }
}  // namespace

// taps of the widest support window of an axis: (int)ceil(max(in / out, 1)) * 2 + 1 (Pillow's ksize)
extern "C" int dir_resize_ksize(int in_size, int out_size) {
    if (in_size <= 0 || out_size <= 0) return 0;
    const double scale = (double)in_size / (double)out_size;
    const double support = scale < 1.0 ? 1.0 : scale;
    int c = (int)support;
    if ((double)c < support) ++c;                                   // ceil
    return c * 2 + 1;
}

extern "C" size_t dir_resize_u8_workspace(int B, int S, int kmax, size_t tmp_bytes) {
    if (B <= 0 || S <= 0 || kmax <= 0) return 0;
    return rz_coef_bytes(B, S, kmax) + dir_align_up(tmp_bytes, 256);
}

extern "C" int dir_resize_u8(const void* src, size_t src_bytes, const long long* table, void* out, int B, int S, int hmax, int kmax, void* workspace,
                             size_t workspace_bytes, dir_stream_t stream) {
    DIR_RETURN_IF(!src || !src_bytes || !table || !out || !workspace || B <= 0 || S <= 0 || hmax <= 0 || kmax <= 0, DIR_EINVAL);
    DIR_RETURN_IF(B > 65535 || (long long)S * S >= (1ll << 31) || (long long)hmax * S >= (1ll << 31), DIR_EUNSUPPORTED);
    const size_t cb = rz_coef_bytes(B, S, kmax);
    DIR_RETURN_IF(workspace_bytes < cb + (size_t)S * 3, DIR_EWORKSPACE);       // (at least one intermediate row behind the coefficient tables)
    const long long tmp_bytes = (long long)(workspace_bytes - cb);
    int* bounds = static_cast<int*>(workspace);
    int* kk = bounds + (size_t)B * 2 * S * 2;
    uint8_t* tmp = static_cast<uint8_t*>(workspace) + cb;            // (the host laid the intermediates out behind cb: table[b][3] are offsets into it)
    hipStream_t s = dir_s(stream);
    hipLaunchKernelGGL(resize_coeffs_kernel, dim3(B, 2), dim3(DIR_TPB), 0, s, table, S, kmax, bounds, kk);
    DIR_LAUNCH_CHECK();
    hipLaunchKernelGGL(resize_h_kernel, dim3(dir_cdiv((long long)hmax * S, DIR_TPB), B), dim3(DIR_TPB), 0, s, static_cast<const uint8_t*>(src), table, S, hmax, kmax,
                       (long long)src_bytes, tmp_bytes, bounds, kk, tmp);
    DIR_LAUNCH_CHECK();
    hipLaunchKernelGGL(resize_v_kernel, dim3(dir_cdiv((long long)S * S, DIR_TPB), B), dim3(DIR_TPB), 0, s, tmp, table, S, hmax, kmax, (long long)src_bytes, tmp_bytes, bounds, kk,
                       static_cast<uint8_t*>(out));
    DIR_LAUNCH_CHECK();
    return DIR_OK;
}
