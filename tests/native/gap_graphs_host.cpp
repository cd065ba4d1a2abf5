// compile check of platanus_b_amd/host/gap_graphs.hpp (the host-side mirror of the batched gap-closing graphs): builds without a GPU, the
// call itself needs one (tests/test_gapclose_graphs.py goes through ctypes).
#include "../../platanus_b_amd/host/gap_graphs.hpp"
int main()
{
    pbh::GapBatch b;
    b.add_sequence(std::vector<char>{0, 1, 2, 3});
    b.end_gap();
    pbh::Graph<1> g;
    pbh::GapNodes r;
    r.node_first = {0, 0};
    r.word_off = {0};
    pbh::gap_graph_tables<1>(g, 21, r, 0);
    return (b.n_gaps() == 1 && g.left.empty()) ? 0 : 1;
}
