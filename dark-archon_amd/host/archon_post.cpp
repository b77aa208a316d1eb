// archon_post.cpp -- optional post-BWT stage of the container CLI (`archon e|d -m -b<size>`): move-to-front, zero-run
// coding and an order-0 canonical Huffman coder per block (BASELINE.json configs[4] "MTF/entropy
// stage").
//
// PARITY UNPINNED: the reference has no such stage -- its README only promises "compression schemes eventually"
// (kvark/dark-archon README.md:2), and a6's Huffman (bwt/a6/src/huff.c) is a PRE-sort key coder, not an output
// coder.  This is a self-consistent host-side stage (round-trip tested only); it is not part of the BWT hot path
// and not a fallback for it: the transform itself still runs on the GPU or fails.
//
// Stream of one piece (the container cuts a block into pieces of 32 KiB, archon_container.cpp):
//   u32 n | u8 code length of each of the 258 symbols | bit stream (LSB first).
// The container's encoder runs this stage on the GPU (csrc/post.hiph, archon_hip_forward_post): same bytes; this file is
// the statement of the format, the decoder, and what tests/test_gpu_post.py holds the device stage against.
// Symbols: 0 = RUNA, 1 = RUNB (a run of r zeros after MTF is written as the bijective base-2 digits of r, as in
// bzip2), 2..256 = MTF value 1..255, 257 = end of block.
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "../../include/archon.h"

namespace {

constexpr int kSyms = 258, kEob = 257, kMaxLen = 20;

// code lengths of an order-0 Huffman code, limited to kMaxLen bits (frequencies are halved until the tree fits)
void huff_lengths(const uint64_t *freq_in, uint8_t *len)
{
    std::vector<uint64_t> freq(freq_in, freq_in + kSyms);
    for (;;) {
        struct Node { uint64_t w; int left, right; };
        std::vector<Node> nodes;
        std::vector<int> live;
        for (int s = 0; s < kSyms; ++s)
            if (freq[s]) { nodes.push_back({freq[s], -1 - s, 0}); live.push_back((int)nodes.size() - 1); }
        memset(len, 0, kSyms);
        if (live.size() == 1) { len[-1 - nodes[live[0]].left] = 1; return; }
        auto cmp = [&](int a, int b) { return nodes[a].w > nodes[b].w || (nodes[a].w == nodes[b].w && a > b); };
        std::make_heap(live.begin(), live.end(), cmp);
        while (live.size() > 1) {
            std::pop_heap(live.begin(), live.end(), cmp); const int a = live.back(); live.pop_back();
            std::pop_heap(live.begin(), live.end(), cmp); const int b = live.back(); live.pop_back();
            nodes.push_back({nodes[a].w + nodes[b].w, a, b});
            live.push_back((int)nodes.size() - 1);
            std::push_heap(live.begin(), live.end(), cmp);
        }
        // depths by walking down from the root (children have smaller indices than their parent)
        std::vector<int> depth(nodes.size(), 0);
        int maxd = 0;
        for (int i = (int)nodes.size() - 1; i >= 0; --i) {
            if (nodes[i].left < 0) { len[-1 - nodes[i].left] = (uint8_t)depth[i]; maxd = std::max(maxd, depth[i]); }
            else { depth[nodes[i].left] = depth[i] + 1; depth[nodes[i].right] = depth[i] + 1; }
        }
        if (maxd <= kMaxLen) return;
        for (int s = 0; s < kSyms; ++s)
            if (freq[s]) freq[s] = (freq[s] + 1) / 2;
    }
}

// canonical codes (shorter codes first, then by symbol), bit-reversed for an LSB-first stream
void canonical(const uint8_t *len, uint32_t *code)
{
    uint32_t next[kMaxLen + 2] = {0}, count[kMaxLen + 2] = {0};
    for (int s = 0; s < kSyms; ++s) ++count[len[s]];
    count[0] = 0;
    uint32_t c = 0;
    for (int l = 1; l <= kMaxLen; ++l) { c = (c + count[l - 1]) << 1; next[l] = c; }
    for (int s = 0; s < kSyms; ++s) {
        if (!len[s]) { code[s] = 0; continue; }
        uint32_t v = next[len[s]]++, r = 0;
        for (int b = 0; b < len[s]; ++b) r |= ((v >> b) & 1u) << (len[s] - 1 - b);
        code[s] = r;
    }
}

struct BitWriter {
    uint8_t *p;
    uint64_t acc = 0;
    int fill = 0;
    void put(uint32_t v, int n)
    {
        acc |= (uint64_t)v << fill;
        fill += n;
        while (fill >= 8) { *p++ = (uint8_t)acc; acc >>= 8; fill -= 8; }
    }
    void flush() { if (fill) { *p++ = (uint8_t)acc; acc = 0; fill = 0; } }
};

// MTF + zero runs -> symbols; `emit(sym)` is called for every symbol (twice: count, then write)
template <class Emit>
void mtf_rle(const uint8_t *bwt, size_t n, Emit emit)
{
    uint8_t order[256];
    for (int i = 0; i < 256; ++i) order[i] = (uint8_t)i;
    uint64_t run = 0;
    auto flush_run = [&]() {
        while (run) {                       // bijective base 2: digits 1 (RUNA) and 2 (RUNB)
            emit((run & 1u) ? 0 : 1);
            run = (run - 1) >> 1;
        }
    };
    for (size_t i = 0; i < n; ++i) {
        const uint8_t c = bwt[i];
        if (order[0] == c) { ++run; continue; }
        flush_run();
        int r = 1;
        while (order[r] != c) ++r;
        memmove(order + 1, order, (size_t)r);
        order[0] = c;
        emit(r + 1);                        // MTF value r in 1..255 -> symbol r + 1
    }
    flush_run();
    emit(kEob);
}

}  // namespace

extern "C" {

// the longest stream n bytes can make: every position one symbol of kMaxLen bits, plus the end symbol (the same figure
// the device stage sizes its piece slots by, csrc/post.hiph)
size_t archon_post_bound(size_t n) { return 4 + (size_t)kSyms + ((n + 1) * kMaxLen + 7) / 8 + 8; }

size_t archon_post_encode(const uint8_t *bwt, size_t n, uint8_t *out)
{
    uint64_t freq[kSyms] = {0};
    mtf_rle(bwt, n, [&](int s) { ++freq[s]; });
    uint8_t len[kSyms];
    uint32_t code[kSyms];
    huff_lengths(freq, len);
    canonical(len, code);
    const uint32_t n32 = (uint32_t)n;
    memcpy(out, &n32, 4);
    memcpy(out + 4, len, kSyms);
    BitWriter bw{out + 4 + kSyms};
    mtf_rle(bwt, n, [&](int s) { bw.put(code[s], len[s]); });
    bw.flush();
    return (size_t)(bw.p - out);
}

int archon_post_decode(const uint8_t *in, size_t in_bytes, uint8_t *bwt, size_t n)
{
    if (in_bytes < 4 + (size_t)kSyms) return -1;
    uint32_t n32;
    memcpy(&n32, in, 4);
    if (n32 != n) return -1;
    const uint8_t *len = in + 4;
    for (int s = 0; s < kSyms; ++s)
        if (len[s] > kMaxLen) return -1;
    uint32_t code[kSyms];
    canonical(len, code);
    // decode table on the low kFast bits; longer codes are matched by a linear scan over the (few) long symbols
    constexpr int kFast = 11;
    std::vector<uint16_t> fast(1u << kFast, 0xFFFF);
    std::vector<int> slow;
    for (int s = 0; s < kSyms; ++s) {
        if (!len[s]) continue;
        if (len[s] <= kFast) {
            for (uint32_t v = code[s]; v < (1u << kFast); v += 1u << len[s]) fast[v] = (uint16_t)s;
        } else {
            slow.push_back(s);
        }
    }
    const uint8_t *p = in + 4 + kSyms, *end = in + in_bytes;
    uint64_t acc = 0;
    int fill = 0;
    uint8_t order[256];
    for (int i = 0; i < 256; ++i) order[i] = (uint8_t)i;
    size_t o = 0;
    uint64_t run = 0, weight = 1;
    for (;;) {
        while (fill <= 56 && p < end) { acc |= (uint64_t)*p++ << fill; fill += 8; }
        int s = fast[acc & ((1u << kFast) - 1)];
        if (s == 0xFFFF) {
            s = -1;
            for (int t : slow)
                if ((acc & ((1ull << len[t]) - 1)) == code[t]) { s = t; break; }
            if (s < 0) return -1;
        }
        if (len[s] > fill) return -1;
        acc >>= len[s];
        fill -= len[s];
        if (s <= 1) {                       // RUNA / RUNB digit
            run += weight << s;             // digit value (s + 1) * weight
            weight <<= 1;
            continue;
        }
        if (run) {
            if (run > n - o) return -1;
            memset(bwt + o, order[0], (size_t)run);
            o += (size_t)run;
            run = 0;
            weight = 1;
        }
        if (s == kEob) break;
        const int r = s - 1;
        const uint8_t c = order[r];
        memmove(order + 1, order, (size_t)r);
        order[0] = c;
        if (o >= n) return -1;
        bwt[o++] = c;
    }
    return o == n ? 0 : -1;
}

}  // extern "C"
