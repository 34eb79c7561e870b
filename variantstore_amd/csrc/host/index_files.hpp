// index_files.hpp -- index directory <-> HostGraph (on-disk drop-in surface).
#pragma once
#include <string>
#include <stdexcept>
#include "host_graph.hpp"
namespace vsamd {
inline void load_index_dir(const std::string&, HostGraph&) { throw std::runtime_error("index directory loader is not built yet"); }
inline void save_index_dir(const HostGraph&, const std::string&) { throw std::runtime_error("index directory writer is not built yet"); }
}
