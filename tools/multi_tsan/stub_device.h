// stub_device.h - what the ThreadSanitizer driver shares with the host stand-in of the device layer (stub_device.cpp).
#pragma once
#include <atomic>
#include <cstdint>

namespace stub {
extern std::atomic<int> fail_shard;
extern std::atomic<uint64_t> fail_code;
extern std::atomic<uint64_t> merges, copies;
extern std::atomic<int> ctx_serial;
uint64_t mix(uint64_t x);
uint32_t fake_count(uint64_t code, uint32_t strand, uint64_t shard_first_word);
uint16_t fake_vote(uint64_t code, uint32_t strand, uint32_t pos);
}  // namespace stub
