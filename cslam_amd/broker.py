#!/usr/bin/env python
"""Communication broker: which matching-graph vertices are shipped between robots.

Drop-in for cslam/broker.py:8-129 (`Broker(edges, robots_involved).brokerage(use_vertex_cover)`),
the consumer of `select_candidates` (gdlcd.py:328-342), written without networkx so that it scales
to the 10^3-10^4 selected edges of the full-loop configuration:
  * components by breadth-first search in vertex insertion order;
  * two robots (bipartite): Hopcroft-Karp maximum matching + the Koenig construction.  The cover
    (L - Z) | (R & Z), Z = vertices reachable by alternating paths from the unmatched L vertices,
    does not depend on which maximum matching was found (Dulmage-Mendelsohn), only on which side
    is called L; networkx takes the side of the first vertex of the component, and so does this;
  * more robots: the local-ratio 2-approximation of Bar-Yehuda & Even with unit weights over the
    edges in insertion order (what networkx `min_weighted_vertex_cover` implements);
  * simple dialog: same draws from `np.random.randint(2)` in the same order as broker.py:111-129.
"""
from collections import deque

import numpy as np


class Broker(object):
    """The broker decides which vertices in the matching
    graph are to be shared between the robots.
    """

    def __init__(self, edges, robots_involved):
        """Initialize the broker

        Args:
            edges (list(EdgeInterRobot)): selected inter-robot edges
            robots_involved (list(int)): Robot ids of the robots involved in the exchange
        """
        self.edges = edges
        involved = set(robots_involved)
        with_edges = set()
        for e in edges:
            if e.robot0_id in involved:
                with_edges.add(e.robot0_id)
            if e.robot1_id in involved:
                with_edges.add(e.robot1_id)
        self.robots_involved_with_edges = sorted(with_edges)
        self.is_multi_robot_graph = len(with_edges) >= 2
        if not self.is_multi_robot_graph:
            return
        self.is_bipartite = len(with_edges) == 2
        # matching graph: vertex = (robot id, keyframe id); adjacency lists keep insertion order
        self.vertex_id = {}
        self.vertices = []
        self.adj = []
        for e in edges:
            v0 = (e.robot0_id, e.robot0_keyframe_id)
            v1 = (e.robot1_id, e.robot1_keyframe_id)
            for v in (v0, v1):
                if v[0] in with_edges and v not in self.vertex_id:
                    self.vertex_id[v] = len(self.vertices)
                    self.vertices.append(v)
                    self.adj.append([])
            if v0[0] in with_edges and v1[0] in with_edges:
                a, b = self.vertex_id[v0], self.vertex_id[v1]
                if b not in self.adj[a]:
                    self.adj[a].append(b)
                    if a != b:
                        self.adj[b].append(a)

    def brokerage(self, use_vertex_cover):
        """Return the broker selection of vertices to send.
        Either using vertex cover or simple dialog strategy.

        Returns:
            List(set((int,int)): Vertices to be transmitted
        """
        if not self.is_multi_robot_graph:
            return []
        return self.vertex_cover() if use_vertex_cover else self.simple_dialog()

    # ------------------------------------------------------------ vertex cover ----
    def components(self):
        """Connected components, each a list of vertex numbers in insertion order."""
        seen = [False] * len(self.vertices)
        out = []
        for s in range(len(self.vertices)):
            if seen[s]:
                continue
            seen[s] = True
            comp, queue = [], deque([s])
            while queue:
                u = queue.popleft()
                comp.append(u)
                for v in self.adj[u]:
                    if not seen[v]:
                        seen[v] = True
                        queue.append(v)
            out.append(sorted(comp))
        return out

    def vertex_cover(self, left_is_first=True):
        """Minimum vertex cover per component (Koenig) for two robots, local-ratio 2-approximation
        otherwise (reference broker.py:83-109).

        Returns:
            List(set((int,int)): Vertices to be transmitted, one set per component
        """
        covers = []
        for comp in self.components():
            if self.is_bipartite:
                ids = self._koenig_cover(comp, left_is_first)
            else:
                ids = self._local_ratio_cover(comp)
            covers.append({self.vertices[i] for i in ids})
        return covers

    def _koenig_cover(self, comp, left_is_first):
        first = next((u for u in comp if self.adj[u]), None)
        if first is None:
            return []
        side = self.vertices[first][0] if left_is_first else \
            next(r for r in self.robots_involved_with_edges if r != self.vertices[first][0])
        left = [u for u in comp if self.vertices[u][0] == side]
        match = self._hopcroft_karp(left)
        # Z: alternating reachability from the unmatched left vertices
        in_z = set(u for u in left if u not in match)
        queue = deque(in_z)
        while queue:
            u = queue.popleft()
            for v in self.adj[u]:                # left -> right over non-matching edges
                if v in in_z or match.get(u) == v:
                    continue
                in_z.add(v)
                w = match.get(v)                 # right -> left over the matching edge
                if w is not None and w not in in_z:
                    in_z.add(w)
                    queue.append(w)
        lset = set(left)
        return [u for u in comp if (u in lset) != (u in in_z)]   # (L - Z) | (R & Z)

    def _hopcroft_karp(self, left):
        """Maximum matching of the bipartite component whose one side is `left`.
        Returns {vertex: partner} for matched vertices of both sides."""
        match = {}
        INF = float("inf")
        while True:
            dist = {}
            queue = deque()
            for u in left:
                if u not in match:
                    dist[u] = 0
                    queue.append(u)
            reach_free = INF
            while queue:
                u = queue.popleft()
                if dist[u] >= reach_free:
                    continue
                for v in self.adj[u]:
                    w = match.get(v)
                    if w is None:
                        reach_free = min(reach_free, dist[u] + 1)
                    elif w not in dist:
                        dist[w] = dist[u] + 1
                        queue.append(w)
            if reach_free == INF:
                return match
            for root in left:
                if root in match:
                    continue
                # iterative depth-first search along the layered graph
                stack = [(root, iter(self.adj[root]))]
                path = []
                while stack:
                    u, it = stack[-1]
                    advanced = False
                    for v in it:
                        w = match.get(v)
                        if w is None:
                            if dist[u] + 1 == reach_free:
                                path.append((u, v))
                                for a, b in path:
                                    match[a] = b
                                    match[b] = a
                                stack = []
                                advanced = True
                                break
                        elif dist.get(w) == dist[u] + 1:
                            path.append((u, v))
                            stack.append((w, iter(self.adj[w])))
                            advanced = True
                            break
                    if not advanced and stack:
                        dist[u] = INF            # dead end in this phase
                        stack.pop()
                        if path:
                            path.pop()

    def _local_ratio_cover(self, comp):
        cost = {u: 1 for u in comp}
        cover = []
        chosen = set()
        done = set()
        for u in comp:
            for v in self.adj[u]:
                if (v, u) in done:
                    continue
                done.add((u, v))
                if u in chosen or v in chosen:
                    continue
                if cost[u] <= cost[v]:
                    chosen.add(u); cover.append(u)
                    cost[v] -= cost[u]
                else:
                    chosen.add(v); cover.append(v)
                    cost[u] -= cost[v]
        return cover

    # ----------------------------------------------------------- simple dialog ----
    def simple_dialog(self):
        """Simple dialog exchange
        For each edge, transmit one of the two vertices randomly
        unless one of the 2 vertices is already transmitted.

        Returns:
            List(set((int,int)): Vertices to be transmitted
        """
        vertices_dialog = set()
        for e in self.edges:
            pair = ((e.robot0_id, e.robot0_keyframe_id), (e.robot1_id, e.robot1_keyframe_id))
            if pair[0] not in vertices_dialog and pair[1] not in vertices_dialog:
                vertices_dialog.add(pair[np.random.randint(2)])
        return [vertices_dialog]
