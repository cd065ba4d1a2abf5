// TEST INFRASTRUCTURE ONLY -- the oracle's graph = the host node-level graph (platanus_b_amd/host/node_graph.hpp,
// which restates graph.h's node tables and cleaning steps) driven by the oracle's own k-mer-level table
// (DHash, dhash.hpp) through BruijnGraph::makeInitialBruijnGraph's DFS (graph.h:289-579).
#pragma once
#include "dhash.hpp"
#include "../../platanus_b_amd/host/node_graph.hpp"
