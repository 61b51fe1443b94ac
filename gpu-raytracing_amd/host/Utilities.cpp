#include "Utilities.h"

#include <cstdio>
#include <vector>

// Both walks use an explicit stack: an LBVH over duplicate-heavy input can be ~60 levels deep, which is fine for
// recursion too, but an SAH top tree handed in by a caller need not be.
HierarchyStats CountNodes(Node* nodes, unsigned root, unsigned count)
{
    HierarchyStats s{0, 0, 0};
    std::vector<uint32_t> st;
    for (unsigned i = count; i-- > 0;)
        if (NodeType(nodes[root + i]) == ChildType_Box) st.push_back(root + i);
    while (!st.empty()) {
        const uint32_t idx = st.back();
        st.pop_back();
        s.numNodes++;
        if (NodeType(nodes[idx]) == ChildType_Tri) s.numLeafNodes++;
        else if (NodeType(nodes[idx]) == ChildType_Box) {
            s.numTreeNodes++;
            for (uint32_t i = NodeCount(nodes[idx]); i-- > 0;) st.push_back(NodeChild(nodes[idx]) + i);
        }
    }
    return s;
}

int VerifyHierarchy(Node* nodes, unsigned root, unsigned count)
{
    int errors = 0;
    std::vector<uint32_t> st;
    for (unsigned i = 0; i < count; i++)
        if (NodeType(nodes[root + i]) == ChildType_Box) st.push_back(root + i);
    while (!st.empty()) {
        const uint32_t idx = st.back();
        st.pop_back();
        if (NodeType(nodes[idx]) != ChildType_Box) continue;
        vec3 lo = make_vec3(FLT_MAX), hi = make_vec3(-FLT_MAX);
        const uint32_t c = NodeChild(nodes[idx]), k = NodeCount(nodes[idx]);
        for (uint32_t i = 0; i < k; i++) {
            lo = fminf(lo, nodes[c + i].min);
            hi = fmaxf(hi, nodes[c + i].max);
        }
        const Node& n = nodes[idx];
        if (n.min.x != lo.x || n.min.y != lo.y || n.min.z != lo.z || n.max.x != hi.x || n.max.y != hi.y || n.max.z != hi.z) {
            fprintf(stderr, "Error: Invalid hierarchy; aabb inclusion check failed on index %d\n", (int)idx);
            errors++;
            continue;  // the reference does not descend below a failing node
        }
        for (uint32_t i = 0; i < k; i++) st.push_back(c + i);
    }
    return errors;
}
