from .branch import BranchSkeleton
from .cloud import Cloud
from .graph import Graph
from .tree import DisjointTreeSkeleton, TreeSkeleton
from .tube import CollatedTube, Tube, collate_tubes

__all__ = ["BranchSkeleton", "Cloud", "Graph", "TreeSkeleton", "DisjointTreeSkeleton", "Tube", "CollatedTube",
           "collate_tubes"]
