// ipk_hash.hpp -- the stage-boundary caching contract of Pipeline::run (src/pipeline.rs:341-372):
//   * BufHasher (src/hasher.rs:12-48): one running hash; `result()` finalises a copy, so hash i covers the
//     settings and every op up to and including op i (a chain: editing op k changes hashes k..7 only);
//   * PipelineCache = MultiCache<BufHash, OpBuffer> (src/pipeline.rs:43,258-260): a byte-budgeted LRU.
//
// The reference feeds bincode-serialised structs to BLAKE3.  Neither the digest function nor the byte
// layout is observable through the reference's API (a BufHash is only ever compared with another BufHash
// made by the same process), so this restates the *contract* -- same fields, same order, same chaining --
// over SHA-256 (FIPS 180-4) and a fixed little-endian layout.  Host-only code.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <list>
#include <map>
#include <memory>
#include <mutex>
#include <array>

namespace ipk {

class Sha256 {
 public:
  Sha256() { reset(); }
  void reset() {
    static const uint32_t iv[8] = {0x6a09e667u, 0xbb67ae85u, 0x3c6ef372u, 0xa54ff53au, 0x510e527fu, 0x9b05688cu, 0x1f83d9abu, 0x5be0cd19u};
    std::memcpy(h_, iv, sizeof(iv)); len_ = 0; fill_ = 0;
  }
  void update(const void *data, size_t n) {
    const uint8_t *p = static_cast<const uint8_t *>(data);
    len_ += n;
    while (n) {
      const size_t take = std::min<size_t>(n, 64 - fill_);
      std::memcpy(block_ + fill_, p, take); fill_ += take; p += take; n -= take;
      if (fill_ == 64) { compress(block_); fill_ = 0; }
    }
  }
  // digest of everything written so far; the running state is left untouched (BufHasher::result, hasher.rs:24-26)
  std::array<uint8_t, 32> result() const {
    Sha256 c = *this;
    const uint64_t bits = c.len_ * 8;
    const uint8_t one = 0x80, zero = 0;
    c.update(&one, 1);
    while (c.fill_ != 56) c.update(&zero, 1);
    uint8_t be[8]; for (int i = 0; i < 8; ++i) be[i] = uint8_t(bits >> (56 - 8 * i));
    c.update(be, 8);
    std::array<uint8_t, 32> out;
    for (int i = 0; i < 8; ++i) for (int b = 0; b < 4; ++b) out[4 * i + b] = uint8_t(c.h_[i] >> (24 - 8 * b));
    return out;
  }

 private:
  static uint32_t rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
  void compress(const uint8_t *b) {
    static const uint32_t K[64] = {
        0x428a2f98u, 0x71374491u, 0xb5c0fbcfu, 0xe9b5dba5u, 0x3956c25bu, 0x59f111f1u, 0x923f82a4u, 0xab1c5ed5u, 0xd807aa98u, 0x12835b01u, 0x243185beu,
        0x550c7dc3u, 0x72be5d74u, 0x80deb1feu, 0x9bdc06a7u, 0xc19bf174u, 0xe49b69c1u, 0xefbe4786u, 0x0fc19dc6u, 0x240ca1ccu, 0x2de92c6fu, 0x4a7484aau,
        0x5cb0a9dcu, 0x76f988dau, 0x983e5152u, 0xa831c66du, 0xb00327c8u, 0xbf597fc7u, 0xc6e00bf3u, 0xd5a79147u, 0x06ca6351u, 0x14292967u, 0x27b70a85u,
        0x2e1b2138u, 0x4d2c6dfcu, 0x53380d13u, 0x650a7354u, 0x766a0abbu, 0x81c2c92eu, 0x92722c85u, 0xa2bfe8a1u, 0xa81a664bu, 0xc24b8b70u, 0xc76c51a3u,
        0xd192e819u, 0xd6990624u, 0xf40e3585u, 0x106aa070u, 0x19a4c116u, 0x1e376c08u, 0x2748774cu, 0x34b0bcb5u, 0x391c0cb3u, 0x4ed8aa4au, 0x5b9cca4fu,
        0x682e6ff3u, 0x748f82eeu, 0x78a5636fu, 0x84c87814u, 0x8cc70208u, 0x90befffau, 0xa4506cebu, 0xbef9a3f7u, 0xc67178f2u};
    uint32_t w[64];
    for (int i = 0; i < 16; ++i) w[i] = (uint32_t(b[4 * i]) << 24) | (uint32_t(b[4 * i + 1]) << 16) | (uint32_t(b[4 * i + 2]) << 8) | b[4 * i + 3];
    for (int i = 16; i < 64; ++i) {
      const uint32_t s0 = rotr(w[i - 15], 7) ^ rotr(w[i - 15], 18) ^ (w[i - 15] >> 3);
      const uint32_t s1 = rotr(w[i - 2], 17) ^ rotr(w[i - 2], 19) ^ (w[i - 2] >> 10);
      w[i] = w[i - 16] + s0 + w[i - 7] + s1;
    }
    uint32_t a = h_[0], bb = h_[1], c = h_[2], d = h_[3], e = h_[4], f = h_[5], g = h_[6], h = h_[7];
    for (int i = 0; i < 64; ++i) {
      const uint32_t t1 = h + (rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25)) + ((e & f) ^ (~e & g)) + K[i] + w[i];
      const uint32_t t2 = (rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22)) + ((a & bb) ^ (a & c) ^ (bb & c));
      h = g; g = f; f = e; e = d + t1; d = c; c = bb; bb = a; a = t1 + t2;
    }
    h_[0] += a; h_[1] += bb; h_[2] += c; h_[3] += d; h_[4] += e; h_[5] += f; h_[6] += g; h_[7] += h;
  }
  uint32_t h_[8]; uint64_t len_; uint8_t block_[64]; size_t fill_;
};

using BufHash = std::array<uint8_t, 32>;

// BufHasher + the serialisation of the op structs: usize -> u64 LE, f32 -> its 4 bytes LE, bool -> 1 byte,
// String / Vec -> u64 length then the elements (the shape bincode's default configuration has).
struct BufHasher {
  Sha256 h;
  void bytes(const void *p, size_t n) { h.update(p, n); }
  void str_raw(const char *s) { h.update(s, std::strlen(s)); }             // hasher.write(self.name().as_bytes()), pipeline.rs:90
  void u64(uint64_t v) { uint8_t b[8]; for (int i = 0; i < 8; ++i) b[i] = uint8_t(v >> (8 * i)); h.update(b, 8); }
  void usize(size_t v) { u64(uint64_t(v)); }
  void boolean(bool v) { const uint8_t b = v ? 1 : 0; h.update(&b, 1); }
  void f32(float v) { uint32_t u; std::memcpy(&u, &v, 4); uint8_t b[4]; for (int i = 0; i < 4; ++i) b[i] = uint8_t(u >> (8 * i)); h.update(b, 4); }
  void f32s(const float *v, size_t n) { for (size_t i = 0; i < n; ++i) f32(v[i]); }
  void string(const char *s) { const size_t n = std::strlen(s); u64(n); h.update(s, n); }
  BufHash result() const { return h.result(); }
};

// MultiCache<BufHash, V> (crate multicache 0.6.0, absent from /root/reference; restated from its documented
// behaviour): entries carry a caller-declared byte size, the total is bounded, `get` refreshes recency, `put`
// evicts least-recently-used entries until the newcomer fits (an entry larger than the whole budget ends up alone).
template <class V>
class LruByteCache {
 public:
  explicit LruByteCache(size_t max_bytes) : max_(max_bytes) {}
  std::shared_ptr<V> get(const BufHash &k) {
    std::lock_guard<std::mutex> lk(mu_);
    auto it = index_.find(k);
    if (it == index_.end()) { ++misses_; return nullptr; }
    order_.splice(order_.end(), order_, it->second);                       // most recently used at the back
    ++hits_;
    return it->second->value;
  }
  bool contains(const BufHash &k) const { std::lock_guard<std::mutex> lk(mu_); return index_.count(k) != 0; }
  void put(const BufHash &k, std::shared_ptr<V> v, size_t bytes) {
    std::lock_guard<std::mutex> lk(mu_);
    auto it = index_.find(k);
    if (it != index_.end()) { total_ -= it->second->bytes; order_.erase(it->second); index_.erase(it); }
    while (total_ + bytes > max_ && !order_.empty()) {
      total_ -= order_.front().bytes; index_.erase(order_.front().key); order_.pop_front(); ++evictions_;
    }
    order_.push_back(Entry{k, std::move(v), bytes});
    index_[k] = std::prev(order_.end());
    total_ += bytes;
  }
  void clear() { std::lock_guard<std::mutex> lk(mu_); order_.clear(); index_.clear(); total_ = 0; }
  void stats(size_t &bytes, size_t &entries, uint64_t &hits, uint64_t &misses, uint64_t &evictions) const {
    std::lock_guard<std::mutex> lk(mu_);
    bytes = total_; entries = order_.size(); hits = hits_; misses = misses_; evictions = evictions_;
  }
  size_t max_bytes() const { return max_; }

 private:
  struct Entry { BufHash key; std::shared_ptr<V> value; size_t bytes; };
  mutable std::mutex mu_;
  std::list<Entry> order_;
  std::map<BufHash, typename std::list<Entry>::iterator> index_;
  size_t max_, total_ = 0;
  uint64_t hits_ = 0, misses_ = 0, evictions_ = 0;
};

}  // namespace ipk
