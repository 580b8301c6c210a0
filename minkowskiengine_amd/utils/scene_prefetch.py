"""Scene prefetch (not in the reference): the coordinate manager of the NEXT scene — hash map, strided maps, kernel
maps, tile plans — is built by a loader thread on its own HIP stream while the training thread launches the current
step.

The reference builds every map lazily inside the forward pass (src/coordinate_map_manager.cpp:763-1018), on the one
stream and the one host thread of the step.  On MI355X a MinkUNet34C step on 200k voxels is ~12 ms of GPU work and the
builds of a new scene ~3 ms of mostly host-bound launches: hidden completely when they run beside the previous step
(bench.py --scenes pipelined).  The native host layer releases the GIL inside insert_and_map / prefetch, so the loader
thread really runs beside the training thread.

    loader = ME.utils.ScenePrefetcher(scene_iterator)       # yields (features, coordinates) on the device
    for x in loader:                                         # x: SparseTensor, maps built, safe on the current stream
        loss = criterion(net(x).F, ...)
        ...
"""
import queue
import threading
import time

import torch


class ScenePrefetcher:
    """Iterate SparseTensors whose coordinate managers were built ahead of time.

    scenes      iterable of (features, coordinates[, kwargs for SparseTensor]) — device tensors; iterated in the loader
                thread under the loader's stream
    depth       scenes built ahead (each holds its maps in device memory)
    The first scene's network requests are recorded lazily (there is no recipe yet); from the second scene on the
    previous manager's recipe is replayed (set_map_prefetch is switched on for the lifetime of the iterator)."""

    def __init__(self, scenes, depth=1, stream=None):
        self._scenes = scenes
        self._depth = max(1, int(depth))
        self._side = stream
        # host-side wall times, one entry per scene: the loader thread's build (insert + replay, including its waits for
        # map sizes) and what the consumer waited for the scene
        self.build_ms, self.wait_ms = [], []

    def __iter__(self):
        from .. import SparseTensor, set_map_prefetch, map_prefetch_enabled, map_prefetch_tag
        from ..coordinate_manager import _prefetch_tag
        tag = getattr(_prefetch_tag, "value", "")     # the consumer's tag travels to the loader thread (thread-local)
        # a HIGH-priority stream: the builds are hundreds of microsecond-sized kernels with a handful of size read-backs
        # between them; behind the training stream's queue of full-chip convolutions each read-back would wait its turn
        side = self._side or torch.cuda.Stream(priority=-1)
        q = queue.Queue(maxsize=self._depth)
        stop = threading.Event()
        was_on = map_prefetch_enabled()
        set_map_prefetch(True)
        device = torch.cuda.current_device()

        def work():
            try:
                torch.cuda.set_device(device)
                _prefetch_tag.value = tag
                it = iter(self._scenes)
                while not stop.is_set():
                    # the iterator itself runs under the loader's stream: host-to-device copies it issues are ordered
                    # before the map builds.  (Tensors it merely hands over must be complete on the device already —
                    # the loader does not wait for the training stream, or nothing would overlap.)
                    t0 = time.perf_counter()
                    with torch.cuda.stream(side):
                        try:
                            item = next(it)
                        except StopIteration:
                            break
                        feats, coords = item[0], item[1]
                        kwargs = item[2] if len(item) > 2 else {}
                        t = SparseTensor(feats, coords, **kwargs)
                        ev = torch.cuda.Event()
                        ev.record(side)
                    self.build_ms.append((time.perf_counter() - t0) * 1e3)
                    q.put((t, ev, (feats, coords)))
                q.put(None)
            except BaseException as e:   # handed to the consumer
                q.put(e)

        th = threading.Thread(target=work, name="me-scene-prefetch", daemon=True)
        th.start()
        try:
            while True:
                t0 = time.perf_counter()
                got = q.get()
                self.wait_ms.append((time.perf_counter() - t0) * 1e3)
                if got is None:
                    break
                if isinstance(got, BaseException):
                    raise got
                t, ev, handed = got
                # Everything the scene holds was allocated (or last written) under the LOADER's stream: the feature
                # matrix (features[unique_index], a segment reduction, a host-to-device copy of the iterator), the
                # coordinates, the unique / inverse maps and the manager's maps and plans.  The consumer's stream — the
                # one that is current HERE, not the one that was current when iteration began — waits for the build and
                # is recorded on every one of them: when the consumer drops the scene while its own kernels are still
                # queued, the caching allocator must not hand those blocks back to the loader's pool for the next
                # scene's build (a cross-stream use-after-free with silent corruption: ADVICE r4).
                cur = torch.cuda.current_stream()
                cur.wait_event(ev)
                t.coordinate_manager.record_stream(cur)
                for x in (t._F, t._C, getattr(t, "unique_index", None), getattr(t, "inverse_mapping", None)) + tuple(handed):
                    if isinstance(x, torch.Tensor) and x.is_cuda:
                        x.record_stream(cur)
                yield t
        finally:
            stop.set()
            while th.is_alive():          # unblock a producer waiting on the full queue
                try:
                    q.get_nowait()
                except queue.Empty:
                    th.join(timeout=0.01)
            set_map_prefetch(was_on)
