// peer.h — internal interface of the peer-mapped scalar exchange (peer.hip), shared with shard.hip.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

namespace mxlo_peer {

constexpr int kMaxPeers = 64;
constexpr int kCap = 4096;                     // doubles per collective (the hot path moves <= 864)
constexpr int kHeaderWords = 8;                // word 0: the sequence number of the payload behind it
constexpr size_t kSlotWords = kHeaderWords + kCap;
// one rank's mailbox: [2 sets][world senders][kSlotWords] 64-bit words
inline size_t mailbox_words(int world) { return 2 * (size_t)world * kSlotWords; }

struct Comm {
  int rank = 0, world = 1, device = 0;
  unsigned long long *mb[kMaxPeers] = {};      // rank r's mailbox as this rank's device addresses it
  unsigned long long seq = 0;                  // collectives LAUNCHED by host calls of this rank (messages only; see seq_dev)
  unsigned long long *seq_dev = nullptr;       // the collective's sequence number lives on the DEVICE: the exchange kernel
                                               // increments and reads it itself, so a hipGraph replay of a captured apply
                                               // advances it like an eager call (a host-side number baked into the kernel
                                               // arguments would make every replay after the first match stale postings)
  unsigned *fault_host = nullptr, *fault_dev = nullptr;
  int timeout_ms = 30000, wall_khz = 100000;
  int drop = -1;                               // TEST HOOK: the rank with this index never posts
  bool dead = false;
};

const char *last_error();
int32_t comm_init_common(Comm *c, int rank, int world, int timeout_ms);
void comm_release_common(Comm *c);
int32_t comm_fault(Comm *c);                   // non-zero (and the comm marked dead) once a gather of this rank timed out
int32_t comm_launch(Comm *c, double *buf, int64_t count, hipStream_t st, int phase);   // phase 1 / 3 advance the device sequence number
int32_t comm_allreduce(Comm *c, double *buf, int64_t count, hipStream_t st);

}  // namespace mxlo_peer
