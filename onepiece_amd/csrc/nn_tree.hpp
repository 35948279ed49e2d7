// nn_tree.hpp -- the host-side search the ICP path re-decides tied queries with (OP_ICP_TIES_REFERENCE): the 3-D instance of
// include/onepiece_nanotree.hpp, which builds the tree the reference's nanoflann would build and descends it the same way, so that
// among exactly equidistant targets the one that library's traversal meets first is returned.
#pragma once
#include "../../include/onepiece_nanotree.hpp"

namespace op_host {
typedef NanoTreeT<3> NanoTree;
}
