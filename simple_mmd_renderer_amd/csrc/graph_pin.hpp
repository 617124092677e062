// graph_pin.hpp -- lifetime bookkeeping between recorded graphs (mmdx_graph_*) and the handles whose device
// buffers they bake in.  A recorded graph holds RAW device addresses of the scratch buffers of every handle that
// took part in the recording (the model; skeletons and motions called with that model).  While such a graph is
// alive the handle is "pinned": a buffer of it that would have to grow (free + malloc) makes the call fail instead,
// and destroying the handle invalidates the graph (mmdx_graph_launch then fails cleanly) -- never a replay into
// freed memory.
#pragma once

#include <atomic>
#include <vector>

struct mmdx_graph_s;
struct mmdx_model_s;

namespace mmdx {

struct GraphPin {
    std::atomic<int> pins{0};             // live graphs that hold addresses of this handle's buffers
    std::vector<mmdx_graph_s *> graphs;   // guarded by the registry's mutex (api.cpp)
    std::vector<mmdx_model_s *> recorders;   // models whose recording IN PROGRESS has used this handle (same mutex)
    std::atomic<bool> replayed{false};    // a graph holding this handle was replayed since the handle last looked (a model's host-side
                                          // record of its morph rates is void then)
};

// A library call is about to enqueue work of `pin`'s handle on `model`'s stream: if that stream is recording, the
// graph being recorded will hold the handle's addresses.
void graph_note_handle(mmdx_model_s *model, GraphPin *pin);
// The handle is going away: every graph that holds its addresses becomes invalid.
void graph_drop_handle(GraphPin *pin);
inline bool graph_pinned(const GraphPin *p) { return p && p->pins.load(std::memory_order_acquire) > 0; }

}  // namespace mmdx
