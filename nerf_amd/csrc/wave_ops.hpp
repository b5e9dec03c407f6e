// 64-lane reductions and prefix scans on the DPP data path (no LDS crossbar round trips): six to seven
// dependent VALU steps per scan instead of the same number of ds_bpermute latencies.
//
// DPP controls used: row_shr:n (0x110 + n) shifts inside a row of 16 lanes; row_bcast:15 (0x142) hands
// lane 15 of each row to the next row, row_bcast:31 (0x143) lane 31 to rows 2 and 3; row_mask / bank_mask
// select the rows / groups of four lanes that take the result, every other lane keeps `old`, which is
// the operation's identity here.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>

namespace nerfhip {

template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ double dpp_f64(double identity, double v) {
    const int ilo = __double2loint(identity), ihi = __double2hiint(identity);
    const int lo = __builtin_amdgcn_update_dpp(ilo, __double2loint(v), CTRL, ROW_MASK, BANK_MASK, false);
    const int hi = __builtin_amdgcn_update_dpp(ihi, __double2hiint(v), CTRL, ROW_MASK, BANK_MASK, false);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double lane_f64(double v, int lane) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane),
                            __builtin_amdgcn_readlane(__double2loint(v), lane));
}

struct OpAdd {
    static constexpr double identity = 0.0;
    __device__ __forceinline__ static double apply(double a, double b) { return a + b; }
};
struct OpMul {
    static constexpr double identity = 1.0;
    __device__ __forceinline__ static double apply(double a, double b) { return a * b; }
};

// inclusive scan over the 64 lanes (lane i receives op over lanes 0..i)
template <typename Op>
__device__ __forceinline__ double wave_scan_f64(double v) {
    constexpr double id = Op::identity;
    double s = Op::apply(v, dpp_f64<0x111, 0xf, 0xf>(id, v));      // row_shr:1
    s = Op::apply(s, dpp_f64<0x112, 0xf, 0xf>(id, v));             // row_shr:2
    s = Op::apply(s, dpp_f64<0x113, 0xf, 0xf>(id, v));             // row_shr:3  -> groups of up to 4
    s = Op::apply(s, dpp_f64<0x114, 0xf, 0xe>(id, s));             // row_shr:4, lanes 4..15 of a row
    s = Op::apply(s, dpp_f64<0x118, 0xf, 0xc>(id, s));             // row_shr:8, lanes 8..15 -> scan inside rows
    s = Op::apply(s, dpp_f64<0x142, 0xa, 0xf>(id, s));             // row_bcast:15 into rows 1 and 3
    s = Op::apply(s, dpp_f64<0x143, 0xc, 0xf>(id, s));             // row_bcast:31 into rows 2 and 3
    return s;
}

// fp32 sum over all 64 lanes, returned in every lane: a balanced tree (row_shr 1/2/4/8 inside rows of 16, then the
// row totals pairwise: rows 1 and 3 take the totals of rows 0 and 2, rows 2 and 3 the sum of rows 0 and 1, so lane 63
// ends with (r3 + r2) + (r1 + r0)), rounding error <= 6 half-ulps of the sum of magnitudes.  A lane without a source
// receives 0.
__device__ __forceinline__ float wave_sum_f32(float v) {
    auto from = [](float x, auto ctrl, auto rows, auto bound) {
        return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), decltype(ctrl)::value, decltype(rows)::value, 0xf,
                                                          decltype(bound)::value));
    };
    using std::integral_constant;
    v += from(v, integral_constant<int, 0x111>{}, integral_constant<int, 0xf>{}, std::true_type{});
    v += from(v, integral_constant<int, 0x112>{}, integral_constant<int, 0xf>{}, std::true_type{});
    v += from(v, integral_constant<int, 0x114>{}, integral_constant<int, 0xf>{}, std::true_type{});
    v += from(v, integral_constant<int, 0x118>{}, integral_constant<int, 0xf>{}, std::true_type{});
    v += from(v, integral_constant<int, 0x142>{}, integral_constant<int, 0xa>{}, std::false_type{});   // row_bcast:15
    v += from(v, integral_constant<int, 0x143>{}, integral_constant<int, 0xc>{}, std::false_type{});   // row_bcast:31
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// op over all 64 lanes, returned in every lane (wave-uniform)
template <typename Op>
__device__ __forceinline__ double wave_reduce_f64(double v) {
    return lane_f64(wave_scan_f64<Op>(v), 63);
}

}  // namespace nerfhip
