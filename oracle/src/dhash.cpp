// TEST INFRASTRUCTURE ONLY -- see dhash.hpp
#include "dhash.hpp"
#include <cmath>

namespace pbo {

u64 table_size_for_memory(u64 memory_bytes, int W)
{
    long base = (long)ref_pair_bytes(W);
    u64 tmp = memory_bytes / base;
    base = std::log(tmp) / std::log(2);
    tmp = std::pow(2, base);
    while (tmp > memory_bytes) tmp >>= 1;
    return tmp;
}

u64 table_size_for_keys(u64 n)
{
    u64 size = std::log(n / 0.9) / std::log(2);
    size = std::pow(2, size + 1);
    return size;
}

} // namespace pbo
