// kmu_hostpack.hpp -- bases cross PCIe at 2 bits (kmu_hostpack.hip): the host-side packer of kmu_sketch_count's pipeline and the
// device kernel that restores the ASCII stream.
#pragma once

#include <vector>

#include "kmu_ctx.hpp"

namespace kmu {

// n ASCII bases -> ceil(n / 4) bytes as Sequence::new(raw, 2) packs them (4 bases per byte, the first in bits 7..6, zero padding);
// false if a byte is outside ACGTacgt (the output is then undefined where that byte went)
bool host_pack2b(const uint8_t *in, uint64_t n, uint8_t *out);

// Packs one base stream on worker threads, slab by slab in order, ahead of whoever uploads it: packed_out[i / 4] holds base i.
class PackPipe {
  public:
    PackPipe(const uint8_t *bases, uint8_t *packed_out, uint64_t total, int threads);
    ~PackPipe();
    bool wait_prefix(uint64_t end); // blocks until bases [0, end) are packed; false once any byte outside ACGTacgt has been met
    PackPipe(const PackPipe &) = delete;
    PackPipe &operator=(const PackPipe &) = delete;

  private:
    struct Impl;
    Impl *im;
};

// n_bases bases from packed_dev (4-byte aligned) to ASCII at out_dev (16-byte aligned), on stream s
int launch_unpack2b(kmu_ctx *ctx, const void *packed_dev, uint64_t n_bases, uint8_t *out_dev, hipStream_t s);

} // namespace kmu
