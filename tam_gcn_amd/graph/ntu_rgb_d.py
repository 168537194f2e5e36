"""NTU-RGB+D 25-joint skeleton graph (reference: graph/ntu_rgb_d.py:7-31).

Parent table, joint 21 (spine-shoulder) is the root.
"""
from . import tools

#            1   2   3  4   5  6  7  8   9 10  11  12 13  14  15  16 17  18  19  20 21  22 23  24  25
_PARENTS = (2, 21, 21, 3, 21, 5, 6, 7, 21, 9, 10, 11, 1, 13, 14, 15, 1, 17, 18, 19, 0, 23, 8, 25, 12)

num_node = len(_PARENTS)
self_link, inward, outward, neighbor = tools.links_from_parents(_PARENTS)


class Graph(tools.SpatialGraph):
    parents = _PARENTS
