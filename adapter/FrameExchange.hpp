// FrameExchange.hpp -- whole-frame hand-off between the compute thread and the display thread.
//
// The reference shares ONE pixel buffer between its compute thread (Main.cpp:96-102) and the GLUT
// thread, which captured the pointer once (SetupGL.cpp:85) and draws from it while the next pass is
// being copied into it (SetupGL.cpp:59-63): tearing is tolerated there.  This is the double-buffered
// `out` of SURVEY 8f-3: three frames in one block -- the writer fills `back()`, publish() makes it the
// latest complete frame, acquire() hands the display the latest complete frame it has not shown yet.
// Neither side ever waits for the other and the reader never sees a frame that is being written: a
// frame is owned by exactly one of {writer, mailbox, reader} at any time (one atomic exchange each).
#ifndef FRAME_EXCHANGE_HPP
#define FRAME_EXCHANGE_HPP

#include <atomic>
#include <cstddef>
#include <cstdint>
#include <vector>

class FrameExchange {
public:
    explicit FrameExchange(size_t pixels) : n(pixels), block(3 * pixels, 0u) {
        seq[0] = seq[1] = seq[2] = 0;
    }
    uint32_t* storage() { return block.data(); }           // the whole block (page-lock it once: rt_pin_output)
    size_t storage_count() const { return block.size(); }

    // writer side (one thread)
    uint32_t* back() { return block.data() + (size_t)w_idx * n; }
    void publish(uint64_t sequence) {                        // `back()` is complete: make it the latest frame
        seq[w_idx] = sequence;
        const int old = mailbox.exchange(w_idx | kFresh, std::memory_order_acq_rel);
        w_idx = old & 3;                                     // whatever the mailbox held is now ours to overwrite
    }

    // reader side (one thread): the latest complete frame; `fresh` says whether it is new since the last call
    const uint32_t* acquire(uint64_t* sequence = nullptr, bool* fresh = nullptr) {
        bool got = false;
        if (mailbox.load(std::memory_order_acquire) & kFresh) {
            const int old = mailbox.exchange(r_idx, std::memory_order_acq_rel);
            r_idx = old & 3;
            got = true;
        }
        if (sequence) *sequence = seq[r_idx];
        if (fresh) *fresh = got;
        return block.data() + (size_t)r_idx * n;
    }

private:
    static constexpr int kFresh = 4;
    const size_t n;
    std::vector<uint32_t> block;
    uint64_t seq[3];
    int w_idx = 0, r_idx = 2;
    std::atomic<int> mailbox{ 1 };
};

#endif
