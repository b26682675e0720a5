"""Sparse matching for loop closure detection on an MI355X.

Drop-in for cslam/loop_closure_sparse_matching.py: same constructor (ROS parameter dict),
same five methods, same public attributes (`local_nnsm`, `other_robots_nnsm`,
`candidate_selector`).  Descriptor banks are HBM-resident `NearestNeighborsMatching`
objects (`ScanContextMatching` for `frontend.sensor_type: lidar`); the candidate bookkeeping is `AlgebraicConnectivityMaximization`.

Batched extensions (`process_local_keyframes`, `process_remote_descriptors`) give the same
matches as calling the per-keyframe reference methods in order, with one GPU launch per
batch instead of one per keyframe (the causal order of
global_descriptor_loop_closure_detection.py:157-160 is kept with a per-row visibility limit).
"""
import numpy as np

from cslam_amd.nns_matching import NearestNeighborsMatching
from cslam_amd.algebraic_connectivity_maximization import AlgebraicConnectivityMaximization, EdgeInterRobot


def _scan_context_matching(device):
    # lidar twin of the descriptor bank (reference lcsm.py:3,21-22,28-29)
    from cslam_amd.lidar_pr.scancontext_matching import ScanContextMatching
    return ScanContextMatching(device=device)


class LoopClosureSparseMatching(object):
    """Sparse matching for loop closure detection
        Matches global descriptors to generate loop closure candidates
        Then candidates are selected such that we respect the communication budget
    """

    def __init__(self, params, device=0):
        """ Initialization of loop closure matching

        Args:
            params (dict): ROS 2 parameters
            device (int): HIP device holding this robot's descriptor banks
        """
        self.params = params
        lidar = self.params["frontend.sensor_type"] == "lidar"
        new_bank = (lambda: _scan_context_matching(device)) if lidar else \
            (lambda: NearestNeighborsMatching(device=device))
        self.local_nnsm = new_bank()
        self.other_robots_nnsm = {}
        for i in range(self.params['max_nb_robots']):
            if i != self.params['robot_id']:
                self.other_robots_nnsm[i] = new_bank()
        self.candidate_selector = AlgebraicConnectivityMaximization(
            self.params['robot_id'], self.params['max_nb_robots'], extra_params=self.params)

    # ------------------------------------------------------------ reference API ----
    def add_local_global_descriptor(self, embedding, keyframe_id):
        """ Add a local keyframe for matching (reference lcsm.py:36-54)

        Args:
            embedding (np.array): global descriptor
            id (int): keyframe id
        """
        matches = []
        self.local_nnsm.add_item(embedding, keyframe_id)
        me = self.params['robot_id']
        for i in range(self.params['max_nb_robots']):
            if i == me:
                continue
            kf, similarity = self.other_robots_nnsm[i].search_best(embedding)
            if kf is not None and similarity >= self.params['frontend.similarity_threshold']:
                match = EdgeInterRobot(me, keyframe_id, i, kf, similarity)
                self.candidate_selector.add_match(match)
                matches.append(match)
        return matches

    def add_other_robot_global_descriptor(self, msg):
        """ Add keyframe global descriptor info from other robot (reference lcsm.py:56-72)

        Args:
            msg (cslam_common_interfaces.msg.GlobalDescriptor): global descriptor info
        """
        descriptor = np.asarray(msg.descriptor)          # float64, as in the reference
        self.other_robots_nnsm[msg.robot_id].add_item(descriptor, msg.keyframe_id)
        kf, similarity = self.local_nnsm.search_best(descriptor)
        if kf is None or not similarity >= self.params['frontend.similarity_threshold']:
            return None
        match = EdgeInterRobot(self.params['robot_id'], kf, msg.robot_id, msg.keyframe_id, similarity)
        self.candidate_selector.add_match(match)
        return match

    @staticmethod
    def _first_valid(kfs, similarities, kf_id, min_gap, threshold):
        for kf, similarity in zip(kfs, similarities):
            if abs(kf - kf_id) < min_gap or similarity < threshold:
                continue
            return kf
        return None

    def match_local_loop_closures(self, descriptor, kf_id):
        """Intra-robot loop closure: best of the top `nb_best_matches` that is far enough in
        time and similar enough (reference lcsm.py:74-92)."""
        kfs, similarities = self.local_nnsm.search(descriptor, k=self.params['frontend.nb_best_matches'])
        if len(kfs) > 0 and kfs[0] == kf_id:
            kfs, similarities = kfs[1:], similarities[1:]
        if len(kfs) == 0 or kfs[0] is None:
            return None, None
        kf = self._first_valid(kfs, similarities, kf_id,
                               self.params['frontend.intra_loop_min_inbetween_keyframes'],
                               self.params['frontend.similarity_threshold'])
        return (kf, kfs) if kf is not None else (None, None)

    def select_candidates(self, number_of_candidates, is_neighbor_in_range, greedy_initialization=True):
        """Select inter-robot loop closure candidates according to budget (reference lcsm.py:94-110)

        Returns:
            list(EdgeInterRobot): selected edges
        """
        return self.candidate_selector.select_candidates(number_of_candidates, is_neighbor_in_range,
                                                         greedy_initialization)

    # ------------------------------------------------------- batched extensions ----
    # Descriptor banks (NearestNeighborsMatching) take the chunk through HBM: it is uploaded once -- or not at all
    # when the extractor hands over its device tensor -- and every add / search of the chunk reads that copy
    # (one chunk feeds 1 add + 1 intra search + one inter search per other robot).  The lidar matcher keeps its
    # host-array interface.
    def _stage(self, embeddings):
        """-> (host array or None, device tensor or None, m).  Query dtype follows the input like the reference
        (float32 stays float32, anything else is float64: nns_matching.py:55-58 through np.dot)."""
        dev_ok = hasattr(self.local_nnsm, "search_device")
        try:
            import torch
        except ImportError:                                     # pragma: no cover
            torch = None
        if torch is not None and isinstance(embeddings, torch.Tensor):
            t = embeddings if embeddings.dtype in (torch.float32, torch.float64) else embeddings.double()
            if dev_ok:
                return None, t.to("cuda:%d" % self.local_nnsm.device).contiguous(), t.shape[0]
            return t.cpu().numpy(), None, t.shape[0]
        emb = np.asarray(embeddings)
        assert emb.ndim == 2
        if dev_ok and emb.shape[0] > 0:
            h = np.ascontiguousarray(emb if emb.dtype == np.float32 else emb.astype(np.float64))
            return emb, torch.from_numpy(h).to("cuda:%d" % self.local_nnsm.device), emb.shape[0]
        return emb, None, emb.shape[0]

    @staticmethod
    def _add(bank, host, dev, ids):
        if dev is not None:
            import torch
            bank.add_items_device(dev if dev.dtype == torch.float32 else dev.float(), ids)   # stored as float32
        else:
            bank.add_items(host, ids)

    @staticmethod
    def _search(bank, host, dev, k, row_limit=None):
        if dev is not None:
            lim = None
            if row_limit is not None:
                import torch
                lim = torch.from_numpy(np.ascontiguousarray(row_limit, dtype=np.int64)).to(dev.device)
            rows, sims, cnt = bank.search_device(dev, int(k), row_limit=lim)
            return rows.cpu().numpy(), sims.cpu().numpy(), cnt.cpu().numpy()
        return bank.search_batch(host, k, row_limit=row_limit)

    def _intra_from_topk(self, rows, sims, cnt, ids, items):
        """match_local_loop_closures' decision (lcsm.py:74-92) for a batch of top-k lists with array operations:
        strip the keyframe itself when it heads its list, then the first entry in score order that is neither closer
        than `intra_loop_min_inbetween_keyframes` nor below the threshold (a NaN similarity is not `< threshold`,
        so it passes, as in the reference).  Returns a list of (kf_id, matched kf or None)."""
        m, k = rows.shape
        gap = self.params['frontend.intra_loop_min_inbetween_keyframes']
        thr = self.params['frontend.similarity_threshold']
        pos = np.arange(k)[None, :]
        valid = pos < cnt[:, None]
        kfs = items[np.where(valid, rows, 0)]
        self_first = valid[:, 0] & (kfs[:, 0] == ids)
        with np.errstate(invalid="ignore"):
            ok = valid & ~(self_first[:, None] & (pos == 0)) & ~(np.abs(kfs - ids[:, None]) < gap) & ~(sims < thr)
        first = ok.argmax(axis=1)
        found = ok[np.arange(m), first]
        hit = kfs[np.arange(m), first].tolist()
        return [(i, (h if f else None)) for i, h, f in zip(ids.tolist(), hit, found.tolist())]

    def process_local_keyframes(self, embeddings, keyframe_ids, intra=True):
        """Batch equivalent of, for each keyframe in order (gdlcd.py:148-174):
               match_local_loop_closures(e, id); add_local_global_descriptor(e, id)
        `embeddings`: [m, d] numpy array or torch tensor (a CUDA tensor is used in place).
        Returns (intra_matches list of (kf_id, matched_kf or None), inter_matches list of
        EdgeInterRobot in the order the sequential calls would produce them)."""
        return self.process_local_keyframes_begin(embeddings, keyframe_ids, intra).finish()

    def process_local_keyframes_begin(self, embeddings, keyframe_ids, intra=True):
        """`process_local_keyframes` in two halves for a host that pipelines steps: this half adds the rows and ENQUEUES the
        searches (local top-k + best-1 per other robot, `cslam_bank_search_multi_enqueue_dev`) without a host
        synchronisation; the returned handle's `finish()` waits for the searches' certificate counts only -- work enqueued in
        between, the next chunk's extraction, keeps running -- and does the host bookkeeping (thresholds, candidate edges).
        Nothing may be added to this robot's banks between the two halves."""
        return _LocalKeyframesStep(self, embeddings, keyframe_ids, intra)

    def _local_keyframes_enqueue(self, embeddings, keyframe_ids, intra):
        host, dev, m = self._stage(embeddings)
        ids = [int(i) for i in keyframe_ids]
        assert len(ids) == m
        n0 = self.local_nnsm.n
        self._add(self.local_nnsm, host, dev, ids)
        me = self.params['robot_id']
        others = [i for i in range(self.params['max_nb_robots']) if i != me and self.other_robots_nnsm[i].n > 0]
        # device banks: the local top-k and the best-1 of every other robot's bank in ONE library call (all kernels
        # enqueued before the host waits for anything, three result copies for the whole chunk)
        pending = None
        if dev is not None and m > 0 and (intra or others):
            import torch
            from cslam_amd.nns_matching import search_multi_device
            k = int(self.params['frontend.nb_best_matches'])
            banks, ks, lims = [], [], []
            if intra:
                lim = torch.arange(n0, n0 + m, dtype=torch.int64, device=dev.device)   # keyframe j sees rows added before it
                banks.append(self.local_nnsm); ks.append(k); lims.append(lim)
            for i in others:
                banks.append(self.other_robots_nnsm[i]); ks.append(1); lims.append(None)
            pending = search_multi_device(banks, dev, ks, lims, defer=True)
        return host, dev, m, ids, n0, others, pending

    def _local_keyframes_finish(self, state, intra):
        host, dev, m, ids, n0, others, pending = state
        results = pending.finish() if pending is not None else None
        ids_arr = np.asarray(ids, dtype=np.int64)
        intra_out = []
        thr = self.params['frontend.similarity_threshold']
        me = self.params['robot_id']
        if intra and m > 0:
            k = int(self.params['frontend.nb_best_matches'])
            if results is not None:
                rows, sims, cnt = results[0]
            else:
                lim = n0 + np.arange(m, dtype=np.int64)       # keyframe j sees rows added before it
                rows, sims, cnt = self._search(self.local_nnsm, host, dev, k, lim)
            item_arr = self.local_nnsm.item_array() if hasattr(self.local_nnsm, "item_array") else None
            if item_arr is not None:
                intra_out = self._intra_from_topk(rows, sims, cnt, ids_arr, item_arr)
            else:                                              # items that are not keyframe numbers (or the lidar bank)
                rows_l, sims_l, cnt_l = rows.tolist(), sims.tolist(), cnt.tolist()
                items = self.local_nnsm.items
                gap = self.params['frontend.intra_loop_min_inbetween_keyframes']
                for j in range(m):
                    c = cnt_l[j]
                    kfs = [items[r] for r in rows_l[j][:c]]
                    s = sims_l[j][:c]
                    if c > 0 and kfs[0] == ids[j]:
                        kfs, s = kfs[1:], s[1:]
                    kf = None
                    if len(kfs) > 0 and kfs[0] is not None:
                        kf = self._first_valid(kfs, s, ids[j], gap, thr)
                    intra_out.append((ids[j], kf))
        inter_out = []
        if others and m > 0:
            # best-1 per other robot, thresholded with array operations; Python only touches the actual matches,
            # in the order the sequential calls produce them (keyframe-major, robot id ascending)
            best_kf = np.empty((m, len(others)), dtype=np.int64)
            best_sims = np.empty((m, len(others)), dtype=np.float64)
            hit = np.zeros((m, len(others)), dtype=bool)
            generic = False
            for c, i in enumerate(others):
                bank = self.other_robots_nnsm[i]
                if results is not None:
                    rows, sims, cnt = results[c + (1 if intra else 0)]
                else:
                    rows, sims, cnt = self._search(bank, host, dev, 1)
                with np.errstate(invalid="ignore"):
                    hit[:, c] = (cnt > 0) & (sims[:, 0] >= thr)
                best_sims[:, c] = sims[:, 0]
                item_arr = bank.item_array() if hasattr(bank, "item_array") else None
                if item_arr is not None:
                    best_kf[:, c] = item_arr[np.where(hit[:, c], rows[:, 0], 0)]
                else:
                    generic = True
                    best_kf[:, c] = rows[:, 0]                 # rows for now, items resolved per match below
            jj, cc = np.nonzero(hit)                           # row-major: j ascending, then robot ascending
            if not generic:
                inter_out = self.candidate_selector.add_matches_arrays(
                    me, ids_arr[jj], np.asarray(others, dtype=np.int64)[cc], best_kf[jj, cc], best_sims[jj, cc])
            else:
                add_match = self.candidate_selector.add_match
                for j, c, r, sv in zip(jj.tolist(), cc.tolist(), best_kf[jj, cc].tolist(), best_sims[jj, cc]):
                    i = others[c]
                    bank = self.other_robots_nnsm[i]
                    kf = r if (hasattr(bank, "item_array") and bank.item_array() is not None) else bank.items[r]
                    match = EdgeInterRobot(me, ids[j], i, kf, sv)
                    add_match(match)
                    inter_out.append(match)
        return intra_out, inter_out

    # (class _LocalKeyframesStep below)

    def process_remote_descriptors(self, robot_id, descriptors, keyframe_ids):
        """Batch equivalent of add_other_robot_global_descriptor for consecutive messages of
        one robot (descriptors as float64 [m, d], like np.asarray(msg.descriptor))."""
        if not hasattr(descriptors, "is_cuda"):
            descriptors = np.asarray(descriptors, dtype=np.float64)
        host, dev, m = self._stage(descriptors)
        if dev is not None and str(dev.dtype) != "torch.float64":
            dev = dev.double()                                   # received descriptors are float64 (lcsm.py:63)
        ids = [int(i) for i in keyframe_ids]
        self._add(self.other_robots_nnsm[robot_id], host, dev, ids)
        out = []
        if self.local_nnsm.n == 0 or m == 0:
            return out
        rows, sims, cnt = self._search(self.local_nnsm, host, dev, 1)
        with np.errstate(invalid="ignore"):
            jj = np.nonzero((cnt > 0) & (sims[:, 0] >= self.params['frontend.similarity_threshold']))[0]
        me = self.params['robot_id']
        item_arr = self.local_nnsm.item_array() if hasattr(self.local_nnsm, "item_array") else None
        if item_arr is not None:
            return self.candidate_selector.add_matches_arrays(
                me, item_arr[rows[jj, 0]], np.full(len(jj), int(robot_id), dtype=np.int64),
                np.asarray(ids, dtype=np.int64)[jj], sims[jj, 0])
        items = self.local_nnsm.items
        for j, r, sv in zip(jj.tolist(), rows[jj, 0].tolist(), sims[jj, 0]):
            match = EdgeInterRobot(me, items[r], robot_id, ids[j], sv)
            self.candidate_selector.add_match(match)
            out.append(match)
        return out

    def process_remote_chunk(self, chunk, last_keyframe_received=-1):
        """One received GlobalDescriptors message in packed form (cslam_amd.wire.DescriptorChunk):
        the body of gdlcd.py:407-422 -- keep the rows newer than the last keyframe received from
        that robot (neighbors_manager.py:147-169), add them, match them.  Returns (matches, updated
        last-received keyframe id)."""
        from cslam_amd.wire import unknown_rows
        rows, last = unknown_rows(chunk, last_keyframe_received)
        if len(rows) == 0:
            return [], last
        desc = chunk.descriptors[rows]
        if hasattr(self.local_nnsm, "search_device"):
            # float32 on the wire -> upload the 4 bytes per value, widen to the receiver's float64 in HBM (exact)
            import torch
            desc = torch.from_numpy(np.ascontiguousarray(desc)).to("cuda:%d" % self.local_nnsm.device).double()
        else:
            desc = desc.astype(np.float64)
        matches = self.process_remote_descriptors(chunk.robot_id, desc, np.asarray(chunk.keyframe_ids)[rows])
        return matches, last

    def process_remote_chunks(self, messages):
        """A drained subscription queue: several received GlobalDescriptors messages at once, [(chunk, last keyframe
        received from that robot), ...] -> [(matches, updated last-received id), ...] in message order.  Same result as
        `process_remote_chunk` message by message (gdlcd.py:407-422 each): every sender's rows go to that sender's bank,
        but the matching against the LOCAL bank -- the same bank for all of them, and untouched by those adds -- is ONE search
        over the concatenated rows instead of one launch + one read-back per message.

        Two messages of the SAME robot in one call behave like two callbacks in a row: the second one is filtered with the
        last id the first one left (neighbors_manager.py:147-169 keeps that state per robot between callbacks), whatever
        `last received` the caller supplied for it -- re-sent or overlapping rows are added and matched once."""
        running = {}                                             # robot id -> last id left by its earlier messages of this call

        def _last_for(chunk, supplied):
            prev = running.get(int(chunk.robot_id))
            return supplied if prev is None else max(int(supplied), prev)

        if not hasattr(self.local_nnsm, "search_device"):
            res = []
            for c, l in messages:
                m, last = self.process_remote_chunk(c, _last_for(c, l))
                running[int(c.robot_id)] = int(last)
                res.append((m, last))
            return res
        import torch
        from cslam_amd.wire import unknown_rows
        devname = "cuda:%d" % self.local_nnsm.device
        staged, out = [], []
        for chunk, last_received in messages:
            rows, last = unknown_rows(chunk, _last_for(chunk, last_received))
            running[int(chunk.robot_id)] = int(last)
            out.append(([], last))
            if len(rows) == 0:
                continue
            dev = torch.from_numpy(np.ascontiguousarray(chunk.descriptors[rows])).to(devname).double()
            ids = np.asarray(chunk.keyframe_ids)[rows].astype(np.int64)
            self._add(self.other_robots_nnsm[chunk.robot_id], None, dev, [int(i) for i in ids])
            staged.append((len(out) - 1, int(chunk.robot_id), ids, dev))
        if not staged or self.local_nnsm.n == 0:
            return out
        rows, sims, cnt = self._search(self.local_nnsm, None, torch.cat([d for _, _, _, d in staged]) if len(staged) > 1 else staged[0][3], 1)
        me = self.params['robot_id']
        thr = self.params['frontend.similarity_threshold']
        item_arr = self.local_nnsm.item_array() if hasattr(self.local_nnsm, "item_array") else None
        items = self.local_nnsm.items if item_arr is None else None
        at = 0
        for slot, robot_id, ids, dev in staged:
            m = len(ids)
            r, sv, c = rows[at:at + m, 0], sims[at:at + m, 0], cnt[at:at + m]
            at += m
            with np.errstate(invalid="ignore"):
                jj = np.nonzero((c > 0) & (sv >= thr))[0]
            if item_arr is not None:
                matches = self.candidate_selector.add_matches_arrays(me, item_arr[r[jj]], np.full(len(jj), robot_id, dtype=np.int64),
                                                                     ids[jj], sv[jj])
            else:
                matches = []
                for j in jj.tolist():
                    match = EdgeInterRobot(me, items[r[j]], robot_id, int(ids[j]), sv[j])
                    self.candidate_selector.add_match(match)
                    matches.append(match)
            out[slot] = (matches, out[slot][1])
        return out


class _LocalKeyframesStep(object):
    """Handle of `LoopClosureSparseMatching.process_local_keyframes_begin`."""

    def __init__(self, lcsm, embeddings, keyframe_ids, intra):
        self._lcsm, self._intra = lcsm, intra
        self._state = lcsm._local_keyframes_enqueue(embeddings, keyframe_ids, intra)
        self._result = None

    def finish(self):
        if self._result is None:
            self._result = self._lcsm._local_keyframes_finish(self._state, self._intra)
            self._state = None
        return self._result
