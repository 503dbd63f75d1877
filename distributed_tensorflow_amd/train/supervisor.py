"""``Supervisor`` and ``Coordinator``: chief election, initialisation, waiting, services, stop.

Reference usage (tfdist_between.py:78, 83, 114; tfdist_between_sync.py:81, 87-89, 120-123)::

    sv = Supervisor(is_chief=(task_index == 0), global_step=global_step, init_op=init_op)
    with sv.prepare_or_wait_for_session(server.target) as sess:
        if task_index == 0: sv.start_queue_runners(sess, [chief_queue_runner])
        ...
        if task_index == 0: sv.request_stop()
    sv.stop()

* chief = task 0.  It creates the session, restores the latest checkpoint from
  ``logdir`` if there is one, otherwise runs ``init_op`` (the initializers
  execute into the parameter-server shards), then marks the variables ready.
* non-chief workers block on that mark.  The wait is event-driven (a blocking
  read on the native server), not TensorFlow's 30-second poll that costs the
  reference's non-chief workers ~37-42 s in their first epoch (README.md:180,
  203, 226, 250-251).
* with a ``logdir`` the chief runs the standard services: a checkpoint thread
  (every ``save_model_secs``) and a final save in ``stop()``.
* ``request_stop()`` / ``stop()`` report the end of this worker's training loop;
  a synchronous chief's queue runner keeps aggregating for the replicas that
  have not finished yet (sized to their number) instead of stranding them at
  the barrier, which is what happens to the reference's non-chief workers when
  the chief leaves first.
* ``stop()`` finally reports the worker as done, which lets the parameter
  servers leave ``join()`` once every worker has finished.
"""
from __future__ import annotations

import contextlib
import os
import threading
import time
from typing import List, Optional

from .. import graph as G
from ..cluster import ConfigProto
from ..session import Session
from .saver import Saver, latest_checkpoint


class Coordinator:
    def __init__(self):
        self._stop = threading.Event()
        self._exc: Optional[BaseException] = None
        self._threads: List[threading.Thread] = []
        self._lock = threading.Lock()

    def request_stop(self, ex: Optional[BaseException] = None):
        with self._lock:
            if ex is not None and self._exc is None:
                self._exc = ex
        self._stop.set()

    def should_stop(self) -> bool:
        return self._stop.is_set()

    def wait_for_stop(self, timeout: Optional[float] = None) -> bool:
        return self._stop.wait(timeout)

    def register_thread(self, t: threading.Thread):
        self._threads.append(t)

    def join(self, threads=None, stop_grace_period_secs: float = 120.0):
        deadline = time.time() + stop_grace_period_secs
        for t in list(threads or self._threads):
            if t.is_alive():
                t.join(max(0.0, deadline - time.time()))
        if self._exc is not None:
            exc, self._exc = self._exc, None
            raise exc

    def clear_stop(self):
        self._stop.clear()


class Supervisor:
    USE_DEFAULT = 0

    def __init__(self, graph: Optional[G.Graph] = None, ready_op=USE_DEFAULT, is_chief: bool = True,
                 init_op=USE_DEFAULT, init_feed_dict=None, local_init_op=USE_DEFAULT, logdir: Optional[str] = None,
                 summary_op=USE_DEFAULT, saver=USE_DEFAULT, global_step=USE_DEFAULT,
                 save_summaries_secs: float = 120, save_model_secs: float = 600, recovery_wait_secs: float = 1,
                 stop_grace_secs: float = 120, checkpoint_basename: str = "model.ckpt", session_manager=None,
                 summary_writer=USE_DEFAULT, init_fn=None):
        self.graph = graph or G.get_default_graph()
        self.is_chief = bool(is_chief)
        self.logdir = logdir
        self.init_op = G.global_variables_initializer() if init_op is Supervisor.USE_DEFAULT else init_op
        self.init_feed_dict = init_feed_dict
        self.init_fn = init_fn
        self.global_step = None if global_step is Supervisor.USE_DEFAULT else global_step
        if self.global_step is None:
            for v in self.graph.variables:
                if v.op_name == "global_step":
                    self.global_step = v
        self.saver = Saver() if saver is Supervisor.USE_DEFAULT else saver
        self.save_model_secs = save_model_secs
        self.recovery_wait_secs = recovery_wait_secs
        self.stop_grace_secs = stop_grace_secs
        self.save_path = os.path.join(logdir, checkpoint_basename) if logdir else None
        self.coord = Coordinator()
        self._session: Optional[Session] = None
        self._threads: List[threading.Thread] = []
        self._restored_from: Optional[str] = None
        self._done_reported = False

    # -- session -----------------------------------------------------------------------
    def prepare_or_wait_for_session(self, master: str = "", config: Optional[ConfigProto] = None,
                                    wait_for_checkpoint: bool = False, max_wait_secs: float = 7200,
                                    start_standard_services: bool = True) -> Session:
        self.coord.clear_stop()
        sess = Session(master, graph=self.graph, config=config)
        if self.is_chief:
            ckpt = latest_checkpoint(self.logdir) if self.logdir else None
            if ckpt is not None and self.saver is not None:
                self.saver.restore(sess, ckpt)
                self._restored_from = ckpt
            else:
                if self.init_op is not None:
                    sess.run(self.init_op, feed_dict=self.init_feed_dict)
                if self.init_fn is not None:
                    self.init_fn(sess)
        else:
            deadline = time.time() + max_wait_secs
            while not sess.store.wait_initialized(self.recovery_wait_secs):
                if time.time() > deadline:
                    raise TimeoutError("variables were not initialised by the chief within max_wait_secs")
            if sess.store.is_remote:
                sess._pull()
        if sess.store.is_remote:
            sess.store.sync_local_step()      # local_step := global_step (TF's local_step_init_op)
            sess._local_step_ready = True
        self._session = sess
        if start_standard_services and self.is_chief and self.logdir:
            self.start_standard_services(sess)
        return sess

    @contextlib.contextmanager
    def managed_session(self, master: str = "", config=None, start_standard_services: bool = True,
                        close_summary_writer: bool = True):
        sess = self.prepare_or_wait_for_session(master, config, start_standard_services=start_standard_services)
        try:
            yield sess
        except Exception as e:
            self.request_stop(e)
            raise
        finally:
            self.stop()
            sess.close()

    # -- services ------------------------------------------------------------------------
    def start_standard_services(self, sess: Session) -> List[threading.Thread]:
        if not (self.is_chief and self.logdir and self.saver is not None and self.save_model_secs):
            return []
        t = threading.Thread(target=self._checkpoint_loop, args=(sess,), name="sv_checkpoint", daemon=True)
        t.start()
        self._threads.append(t)
        self.coord.register_thread(t)
        return [t]

    def _checkpoint_loop(self, sess: Session):
        sess.bind_thread()       # (this thread's launches and copies belong on the session's GPU, not on device 0)
        while not self.coord.wait_for_stop(self.save_model_secs):
            try:
                self.saver.save(sess, self.save_path, global_step=self._step_value(sess))
            except Exception as e:  # pragma: no cover
                print(f"[supervisor] checkpoint failed: {e}", flush=True)

    def _step_value(self, sess: Session) -> int:
        return sess.store.global_step() if self.global_step is not None else 0

    def start_queue_runners(self, sess: Session, queue_runners=None) -> List[threading.Thread]:
        threads = []
        for qr in queue_runners or []:
            for t in qr.create_threads(sess, coord=self.coord, daemon=True, start=True):
                threads.append(t)
                self._threads.append(t)
                self.coord.register_thread(t)
        return threads

    # -- stop -------------------------------------------------------------------------------
    def request_stop(self, ex=None):
        self._report_finished()
        self.coord.request_stop(ex)

    def _report_finished(self):
        sess = self._session
        if sess is not None and sess.server is not None and sess.store.is_remote:
            try:
                sess.server.report_training_finished()
            except Exception:
                pass

    def should_stop(self) -> bool:
        return self.coord.should_stop()

    def wait_for_stop(self):
        self.coord.wait_for_stop()

    def stop(self, threads=None, close_summary_writer: bool = True):
        self._report_finished()
        self.coord.request_stop()
        try:
            self.coord.join(threads or self._threads, stop_grace_period_secs=self.stop_grace_secs)
        finally:
            sess = self._session
            if sess is not None and self.is_chief and self.logdir and self.saver is not None:
                try:
                    self.saver.save(sess, self.save_path, global_step=self._step_value(sess))
                except Exception as e:  # pragma: no cover
                    print(f"[supervisor] final checkpoint failed: {e}", flush=True)
            if sess is not None and sess.server is not None and not self._done_reported:
                self._done_reported = True
                try:
                    sess.server.report_worker_done()
                except Exception:
                    pass
            self._threads = []
