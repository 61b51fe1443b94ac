// Utilities.h -- mirrors the reference's Utilities.h:3-10 (Utilities.cpp:8-83): the reference's only correctness check.
#pragma once
#include "Common.h"

struct HierarchyStats { int numNodes, numLeafNodes, numTreeNodes; };

HierarchyStats CountNodes(Node* nodes, unsigned root, unsigned count);
// prints "Error: Invalid hierarchy; aabb inclusion check failed on index N" per failing node (as the reference);
// additionally returns the number of failures.
int VerifyHierarchy(Node* nodes, unsigned root, unsigned count);
