// TEST INFRASTRUCTURE.  Pins what the reference's scalar_t = at::Half arithmetic computes.
//
// /root/reference/nerf/gridencoder/src/gridencoder.cu is CUDA-only and cannot be compiled here, but the
// meaning of its at::Half expressions is decided by c10::Half's operator set, which ships header-only with
// this image's torch (torch/headeronly/util/Half.h; the same operators are used on host and device).  This
// file evaluates expressions of the reference's shape -- acc += w * g (gridencoder.cu:187),
// acc += w * (r - l) * pd (:235), acc += g * d (:364), (half)(w * g) followed by a half add (:325-331) --
// with c10::Half operands, so that overload resolution and rounding are the real header's.
// tests/test_oracle_grid.py compiles it with g++ and compares with oracle/grid_oracle.c's spelled-out
// roundings, bit for bit.
#include <cstdint>
#include <cstring>

#include <torch/headeronly/util/Half.h>

using c10::Half;

static Half from_bits(uint16_t b) { return Half(b, Half::from_bits()); }

extern "C" {
uint16_t pin_forward_step(uint16_t acc_bits, float w, uint16_t g_bits) {
    Half results = from_bits(acc_bits);
    const Half grid = from_bits(g_bits);
    results += w * grid;
    return results.x;
}
uint16_t pin_jacobian_step(uint16_t acc_bits, float w, uint16_t right_bits, uint16_t left_bits, float pos_deriv) {
    Half results_grad = from_bits(acc_bits);
    const Half r = from_bits(right_bits), l = from_bits(left_bits);
    results_grad += w * (r - l) * pos_deriv;
    return results_grad.x;
}
uint16_t pin_input_backward_step(uint16_t acc_bits, uint16_t grad_bits, uint16_t jac_bits) {
    Half result = from_bits(acc_bits);
    result += from_bits(grad_bits) * from_bits(jac_bits);
    return result.x;
}
uint16_t pin_table_backward_step(uint16_t row_bits, float w, uint16_t grad_bits) {
    const Half v = (Half)(w * from_bits(grad_bits));     // the __half2 lane of gridencoder.cu:329
    Half row = from_bits(row_bits);
    row = row + v;                                        // a half atomic add: float add, one rounding
    return row.x;
}
}
