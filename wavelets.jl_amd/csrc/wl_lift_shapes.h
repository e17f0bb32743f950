// wl_lift_shapes.h -- lifting scheme shapes known at compile time, shared by wl_lift.hip and wl_lift_tile.hip
#pragma once
#include "wl_internal.h"

namespace wl {

// ---- scheme shapes known at compile time (coefficients stay run-time data) -----------------------
// direction-adjusted order (as produced by make_scheme): step i = {is_update, nc, shift}
struct StepShape { int upd, nc, sh; };
// the shape-specialised kernels never have more steps than this; their argument blocks carry only these coefficients
// (keeping kernel arguments small matters: launches with > 256 bytes of arguments were seen to stall the enqueue path)
constexpr int LIFT_FAST_STEPS = 4;
template <int ID> struct Shape;
// cdf9/7 forward and inverse have the same shape sequence read in opposite order
template <> struct Shape<0> { static constexpr int NS = 4; static constexpr StepShape S[4] = {{1, 2, 0}, {0, 2, 1}, {1, 2, 0}, {0, 2, 1}}; };   // cdf9/7 fw
template <> struct Shape<1> { static constexpr int NS = 4; static constexpr StepShape S[4] = {{0, 2, 1}, {1, 2, 0}, {0, 2, 1}, {1, 2, 0}}; };   // cdf9/7 inv
template <> struct Shape<2> { static constexpr int NS = 3; static constexpr StepShape S[3] = {{0, 1, 0}, {1, 2, 1}, {0, 1, -1}}; };             // db2 fw
template <> struct Shape<3> { static constexpr int NS = 3; static constexpr StepShape S[3] = {{0, 1, -1}, {1, 2, 1}, {0, 1, 0}}; };             // db2 inv
template <> struct Shape<4> { static constexpr int NS = 2; static constexpr StepShape S[2] = {{0, 1, 0}, {1, 1, 0}}; };                          // haar/db1 fw
template <> struct Shape<5> { static constexpr int NS = 2; static constexpr StepShape S[2] = {{1, 1, 0}, {0, 1, 0}}; };                          // haar/db1 inv

// dependency cone of a scheme in (s, d) pairs: how far a pair's final value reaches to the left / right
template <int ID>
struct LiftReach {
    static constexpr int left()
    {
        int v = 0;
        for (int k = 0; k < Shape<ID>::NS; ++k) { const int a = Shape<ID>::S[k].sh; if (a > 0) v += a; }
        return v;
    }
    static constexpr int right()
    {
        int v = 0;
        for (int k = 0; k < Shape<ID>::NS; ++k) { const int b = Shape<ID>::S[k].nc - 1 - Shape<ID>::S[k].sh; if (b > 0) v += b; }
        return v;
    }
    static constexpr int HP = left() > right() ? left() : right();
};


}  // namespace wl
