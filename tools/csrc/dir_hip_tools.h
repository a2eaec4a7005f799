/* libdir_hip_tools.so — measurement and test probes that are NOT part of the product library (libdir_hip.so, include/dir_hip.h):
 * box-calibration kernels that bench.py's `peaks` leg and the tools/ scripts time, and the lane-mapping probe of the transposing
 * LDS read. Nothing of the reference's path is here; only bench.py (peaks), tools/ and tests/ load this library. */
#ifndef DIR_HIP_TOOLS_H
#define DIR_HIP_TOOLS_H
#include <stddef.h>
#include "dir_hip.h"
#ifdef __cplusplus
extern "C" {
#endif
/* Box calibration (a measured STREAM-style HBM number and a measured MFMA peak of THIS box, reported next to the
 * nominal 8 TB/s / 2.5 PFLOP/s).
 *   dir_probe_stream_copy: dst[i] = src[i], 16 B per lane, 2048 workgroups grid-stride  (moves 2 * bytes)
 *   dir_probe_stream_read: read-only stream, wave-reduced, out >= 8192 floats           (moves bytes)
 *   dir_probe_stream_write: write-only stream; bytes % 16384 == 0                       (moves bytes)
 *   dir_probe_mfma_bf16 / _f32: `iters` rounds of independent MFMA chains per wavefront, no memory traffic;
 *                          *flops (host, nullable) receives the FLOPs of the launch; out >= workgroups * 256 floats. */
int dir_probe_stream_copy(const void* src, void* dst, size_t bytes, dir_stream_t stream);
int dir_probe_stream_read(const void* src, float* out, size_t bytes, dir_stream_t stream);
int dir_probe_stream_write(void* dst, size_t bytes, dir_stream_t stream);
int dir_probe_mfma_bf16(int workgroups, int iters, float* out, double* flops, dir_stream_t stream);
int dir_probe_mfma_f32(int workgroups, int iters, float* out, double* flops, dir_stream_t stream);
/* L2-resident re-read probe: `workgroups` workgroups each stream the same region_bytes (2-8 MB: L2 resident) `passes` times with
 * `depth` (4 | 8 | 16) independent 16-byte loads per lane in flight: the L2 -> CU delivery rate when the latency is covered.
 * out: >= 4 * workgroups floats. */
int dir_probe_l2_read(const void* src, float* out, size_t region_bytes, int workgroups, int passes, int depth, dir_stream_t stream);
/* LDS holds the uint16 ramp 0, 1, 2, ... (8192 elements); lane l of ONE wavefront issues ds_read_b64_tr_b16 at byte address
 * addr_bytes[l] (8-byte aligned) and out[4 l .. 4 l + 3] receives its four 16-bit results. */
int dir_probe_tr16(const int* addr_bytes, void* out, dir_stream_t stream);
#ifdef __cplusplus
}
#endif
#endif
