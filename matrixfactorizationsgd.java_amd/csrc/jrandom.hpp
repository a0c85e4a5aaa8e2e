// jrandom.hpp -- java.util.Random (JDK specification: 48-bit LCG) with O(log n)
// skip-ahead, so that a Java host and this library seed factors identically
// and a DSGD rank can seed just its own rows.
#pragma once

#include <cstdint>

namespace mfsgd {

class JRandom {
   public:
    explicit JRandom(int64_t seed) : s_((static_cast<uint64_t>(seed) ^ kMult) & kMask) {}

    int32_t next(int bits) {
        s_ = (s_ * kMult + kAdd) & kMask;
        return static_cast<int32_t>(static_cast<uint32_t>(s_ >> (48 - bits)));
    }
    int32_t nextInt() { return next(32); }
    float nextFloat() { return static_cast<float>(next(24)) / static_cast<float>(1 << 24); }

    // Advance the generator by n calls of next().
    void skip(uint64_t n) {
        uint64_t acc_a = 1, acc_c = 0, cur_a = kMult, cur_c = kAdd;
        while (n) {
            if (n & 1) {
                acc_a = (acc_a * cur_a) & kMask;
                acc_c = (acc_c * cur_a + cur_c) & kMask;
            }
            cur_c = ((cur_a + 1) * cur_c) & kMask;
            cur_a = (cur_a * cur_a) & kMask;
            n >>= 1;
        }
        s_ = (acc_a * s_ + acc_c) & kMask;
    }

   private:
    static constexpr uint64_t kMult = 0x5DEECE66DULL;
    static constexpr uint64_t kAdd = 0xBULL;
    static constexpr uint64_t kMask = (1ULL << 48) - 1;
    uint64_t s_;
};

}  // namespace mfsgd
