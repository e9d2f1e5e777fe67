#pragma once
#include "common.h"
#include <algorithm>

// zero / zero_bytes: a range the same launch sets to zero (16-byte granules), or null
int nchw_f32_to_nhwc_f16(const float* src, half_t* dst, int N, int C, int HW, int Cpad, hipStream_t s, void* zero = nullptr, size_t zero_bytes = 0);
int nhwc_f16_to_nchw(const half_t* src, void* dst, int out_f32, int N, int C, int HW, int ld, hipStream_t s);
int nchw_to_nhwc_f16_scaled(const void* src, int src_f32, half_t* dst, int N, int C, int HW, int ld, float mul, hipStream_t s);
int concat2(const half_t* a, const half_t* b, half_t* o, long long M, int Ca, int Cb, hipStream_t s,
            const long long* sa = nullptr, const long long* sb = nullptr, long long* so = nullptr, int N = 1);
int add_f16(const half_t* a, const half_t* b, half_t* o, long long n, hipStream_t s);
struct TsArg { float t[16]; };   // timesteps by value: no host->device copy on the step path
int timestep_embedding(const TsArg& t, float* out, int N, int dim, hipStream_t s);
int gemv_f32(const float* W, const float* b, const float* in, float* out, int rows, int K, int N, int silu_in, hipStream_t s);
// dst_ld / col_off: destination row stride (0 = taps*cpad) and first column -- lets two weights share one K-concatenated matrix
int pack_conv_weight(const float* w, half_t* dst, int O, int I, int taps, int rows_pad, int cpad, int transpose_flip, hipStream_t s,
                     int dst_ld = 0, int col_off = 0);
int pack_conv_weight_split(const float* w, half_t* dst, int O, int I, int taps, int rows_pad, hipStream_t s);
int round_through_f16(const float* src, float* dst, long long n, hipStream_t s);
// attention backward pieces + gradient export (attn_bwd.hip)
int nhwc_f16_to_nchw_f32_scaled(const half_t* src, float* dst, int N, int C, int HW, int ld, const float* mul_dev, hipStream_t s);
