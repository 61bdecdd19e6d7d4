// seal/util/uintarith.h -- present because MOAI's include/source/ckks_evaluator.h:4 includes it by name and calls
// exponentiate_uint (:84).  The rest of SEAL/util/uintarith.h (multi-word integer helpers) is host utility code outside
// the evaluator path and unused by MOAI's callers.
#pragma once
#include <cstdint>

#include "seal/seal.h"

namespace seal
{
    namespace util
    {
        // operand^exponent modulo 2^64 by square and multiply (SEAL/util/uintarith.h:1001, uintarith.cpp:365-397)
        inline std::uint64_t exponentiate_uint(std::uint64_t operand, std::uint64_t exponent)
        {
            if (operand == 0 && exponent == 0)
            {
                throw std::invalid_argument("undefined operation");
            }
            std::uint64_t result = 1;
            while (exponent)
            {
                if (exponent & 1)
                {
                    result *= operand;
                }
                operand *= operand;
                exponent >>= 1;
            }
            return result;
        }
    } // namespace util
} // namespace seal
