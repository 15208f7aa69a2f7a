// The ORDER OF OPERATIONS of a shard's enqueue-only passes (gk_table_sweep_sharded with GK_SHARD_ENQUEUE):
// which stream runs what, which event guards which slot buffer.  One definition, two backends --
//   kernels.hip             HIP streams / events, RCCL's all-gather: the product
//   tests/native/hostemu.cpp  the TEST-ONLY CPU build: a worker thread for the exchange stream, condition variables for events,
//                           the collective a callback (gloo) -- so that the OVERLAPPED exchange, which needs a second rank to mean
//                           anything, is executed at world size 2..4 in the GPU-less container (tests/test_sweep_dist.py), pass by
//                           pass against the single-process answer, before it ever meets a second GPU.
// A pass = sweep (the dominant kernel over the local shard, into this rank's slot of a [world][slot] buffer) + tail (the slot's
// fail-closed counts) + exchange (ONE in-place all-gather of the slots + the totals over the gathered tails).
//   serial      everything on the table's stream, one slot buffer.
//   overlapped  two slot buffers: sweep + tail of pass k+1 fill the other buffer on the table's stream while the exchange of pass k
//               runs on the exchange stream.  EV_SWEEP[s]: buffer s is complete (the exchange may read it); EV_XCHG[s]: the exchange
//               of buffer s is over (a later pass may overwrite it).  Whoever reads an answer outside the passes drains first.
#pragma once

namespace gk {

enum ShardStream { SS_TABLE = 0, SS_COMM = 1 };
enum ShardEvent { SE_SWEEP = 0, SE_XCHG = 1 };

// B: select(s) -- results and slot pointers now mean buffer s | sweep() | tail() -- both on the table's stream | gather(stream) |
//    totals(stream) | record(event, s, stream) | wait(stream, event, s) | sync(stream)
template <class B>
struct ShardPipe {
  bool overlap = false;
  int cur = 0;                        // slot buffer of the most recent pass
  bool pending[2] = {false, false};   // an exchange of buffer b may still be running on the exchange stream

  void enqueue(B& b) {
    if (!overlap) {
      b.select(cur);
      b.sweep(); b.tail();
      b.gather(SS_TABLE); b.totals(SS_TABLE);
      return;
    }
    const int s = cur ^ 1;
    if (pending[s]) b.wait(SS_TABLE, SE_XCHG, s);   // the buffer is reused only after its exchange finished
    cur = s;
    b.select(s);
    b.sweep(); b.tail();
    b.record(SE_SWEEP, s, SS_TABLE);
    b.wait(SS_COMM, SE_SWEEP, s);
    b.gather(SS_COMM); b.totals(SS_COMM);
    b.record(SE_XCHG, s, SS_COMM);
    pending[s] = true;
  }
  // before a collecting sweep rewrites the current buffer, or an answer is downloaded
  void drain(B& b) {
    if (!overlap) return;
    b.sync(SS_COMM);
    pending[0] = pending[1] = false;
  }
  void reset() { cur = 0; pending[0] = pending[1] = false; }
};

}  // namespace gk
