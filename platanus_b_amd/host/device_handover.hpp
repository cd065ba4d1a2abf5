// The one hand-over of the device's graph stage to the host replay, used by `assemble` and `merge` alike:
// pb_graph_adjacency (arrays stay on the device) -> pb_graph_chains -> pb_graph_nodes -> fetch of the chain records and of the
// nodes the DFS can land on -> make_initial_from_nodes (device_graph.hpp), which checks every device-produced index before it
// uses it.  (Round 2's `merge` still took the per-k-mer arrays and make_initial_from_chains; the one unexplained SIGSEGV of that
// round was a `merge` run -- DESIGN.md section 7.)
#pragma once
#include "device_graph.hpp"
#include <functional>

extern "C" {
#include "../../include/pbgpu.h"
}

namespace pbh {

// `chk` turns a libpbgpu status into the caller's exception; `lap(name)` (optional) is called before each stage for stage clocks
template <int W>
void graph_from_device(pb_ctx *ctx, Graph<W> &g, unsigned kk, const std::function<void(int)> &chk, const std::function<void(const char *)> &lap = nullptr)
{
    NodeSet<W> nd;
    ChainRecords cr;
    nd.k = kk;
    if (lap) lap("adjacency (device)");
    chk(pb_graph_adjacency(ctx, nullptr, nullptr));
    if (lap) lap("chain compaction + node export (device + D2H)");
    uint64_t nc = 0, words = 0, nn = 0;
    chk(pb_graph_chains(ctx, &nc, &words));
    chk(pb_graph_nodes(ctx, &nn));
    cr.len.resize(nc); cr.end_a.resize(nc); cr.end_b.resize(nc); cr.root.resize(nc); cr.sum.resize(nc); cr.seq_woff.resize(nc); cr.seq.resize(words);
    chk(pb_graph_fetch_chain_records(ctx, cr.len.data(), (uint64_t *)cr.sum.data(), cr.end_a.data(), cr.end_b.data(), cr.root.data(), (uint64_t *)cr.seq_woff.data(),
                                     (uint64_t *)cr.seq.data()));
    nd.idx.resize(nn); nd.meta.resize(nn); nd.chain.resize(nn); nd.nbr.resize(2 * nn); nd.nb8.resize(8 * nn); nd.keys.resize(nn * W);
    chk(pb_graph_fetch_nodes(ctx, nd.idx.data(), nd.meta.data(), nd.chain.data(), nd.nbr.data(), nd.nb8.data(), (uint64_t *)nd.keys.data()));
    if (lap) lap("DFS replay, chain level (host)");
    make_initial_from_nodes(g, nd, cr);
}

} // namespace pbh
