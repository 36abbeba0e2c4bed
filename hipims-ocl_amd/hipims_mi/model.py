"""Orchestrator above the C ABI for ONE Cartesian domain: what `CModel::runModelMain` and
`CSchemeGodunov::runSimulation / Threaded_runBatch` do around the device work -- sync targets clipped to the output
interval, the batch-size autotuner, skipped-iteration handling at sync points, progress / rate reporting and the
output rasters -- restated for a single domain.

    SchemeDriver  ~ CSchemeGodunov's host-side control  (src/Schemes/CSchemeGodunov.cpp:1053-1092, :1147-1453,
                    :1568-1612, :1741-1835)
    Model         ~ CModel::runModelMain for one local domain (src/CModel.cpp:536-698, :723-891, :905-1100)

Note on reproducibility: in the reference the batch size comes from wall-clock timing, and batch boundaries are not
physics-neutral there -- with quirk Q1 the timestep after a sync point is priced on the pre- or post-step buffer
depending on the parity of the iteration that landed on the sync point and on whether skipped iterations followed.
This restatement keeps the mechanism; pass a fixed batch size (`queueMode="fixed"` in the reference) for
repeatable runs (tests/test_frontend.py shows both).

Multi-domain links, forecast/rollback and MPI are deliberately absent: the strip runner (strips.py) replaces them.
The engine is any object with the `Domain` surface; the CPU tests pass an oracle-backed one.
"""
from __future__ import annotations

import math
import os
import time

from . import frontend


def seconds_to_time(s: float) -> str:
    """Util::secondsToTime: d/h/m/s text of the progress block (src/util.cpp:130-172 shape, not byte-for-byte)."""
    s = max(0.0, float(s))
    if s < 60:
        return f"{s:.2f}s" if s < 10 else f"{s:.1f}s"
    d, rem = divmod(int(s), 86400)
    h, rem = divmod(rem, 3600)
    m, sec = divmod(rem, 60)
    out = (f"{d}d " if d else "") + f"{h:02d}:{m:02d}:{sec:02d}"
    return out


class SchemeDriver:
    """Host-side state of one scheme: the part of CSchemeGodunov that decides WHAT to queue."""

    def __init__(self, sim, cfg, cells):
        self.sim, self.cfg = sim, cfg
        self.cells = cells
        # CScheme.cpp:46-55, CSchemeGodunov.cpp:42-75 defaults; single domain: rollback limit "infinite"
        # (CDomainBase.cpp:163-174)
        self.automatic_queue = True
        self.queue_addition_size = 1
        self.rollback_limit = 999999999
        self.running = False
        self.prepare_simulation()

    # ---- adapters: Domain (HIP engine) and OracleSim spell a few things differently ----
    def _call(self, *names):
        for n in names:
            f = getattr(self.sim, n, None)
            if f is not None:
                return f
        raise AttributeError(names[0])

    def _scalars(self):
        s = self._call("read_scalars", "scalars")()
        if "time" in s:
            return s["time"], s["timestep"], s["batch_timesteps"], s["batch_successful"], s["batch_skipped"]
        return s["t"], s["dt"], s["batch_dt"], s["batch_ok"], s["batch_skipped"]

    # CSchemeGodunov::prepareSimulation (:1053-1092)
    def prepare_simulation(self):
        self.target_time = 0.0
        self.current_time = 0.0
        self.current_timestep = self.cfg.timestep
        self.batch_timesteps = 0.0
        self.batch_successful = self.batch_skipped = 0
        self.batch_rate = 1
        self.batch_started = 0.0
        self.iterations_since_sync = 0
        self.cells_calculated = 0
        self.iterations = 0
        self.update_target = False
        self.override_timestep = False
        self.use_forced_time_advance = True
        self.read_key_statistics()

    # :1741-1753
    def set_target_time(self, t):
        if t == self.target_time:
            return
        self.target_time = t
        self.update_target = True

    def force_time_advance(self):
        self.use_forced_time_advance = True

    # :1817-1835
    def read_key_statistics(self):
        last = self.batch_successful
        (self.current_time, self.current_timestep, self.batch_timesteps, self.batch_successful,
         self.batch_skipped) = self._scalars()
        self.batch_rate = (self.batch_successful - last) if self.batch_successful > last else 1

    # :1568-1612 (forecast sync, one domain)
    def is_sync_ready(self, expected_target):
        return not self.running and not (expected_target - self.current_time > 1e-5)

    def is_suspended(self):
        return self.current_timestep < 0.0

    # :1374-1453 followed by one pass of Threaded_runBatch (:1147-1372); blocking here, where the reference hands the
    # pass to its worker thread and polls isRunning()
    def run_simulation(self, target, real_time):
        if self.running:
            return
        if self.target_time != target:
            self.set_target_time(target)
        if target <= 0.0:
            return
        if self.current_time > target + 1e-5:                      # :1389-1407 (a warning in the reference)
            return
        if self.automatic_queue and real_time > 1e-5:              # :1420-1448, single-domain branch
            duration = real_time - self.batch_started
            old = self.queue_addition_size
            if duration > 0:
                want = int(math.ceil(1.0 / (duration / float(old))))
                self.queue_addition_size = max(1, min(self.batch_rate * 3, want))
            if self.queue_addition_size > old * 2 and self.queue_addition_size > 40:
                self.queue_addition_size = min(self.batch_rate * 3, old * 2)
            self.queue_addition_size = min(self.queue_addition_size, self.rollback_limit - self.iterations_since_sync)
            self.queue_addition_size = max(1, self.queue_addition_size)
        self.batch_started = real_time
        self.running = True
        try:
            if self.update_target:                                 # :1163-1209
                self.update_target = False
                self._call("set_target_time", "set_target")(self.target_time)
                self.iterations_since_sync = 0
                self.use_forced_time_advance = True
                if self.current_timestep <= 0.0:                   # a suspended scheme needs a new timestep: queued on the
                    self._call("update_timestep")()                # device; the host copy is NOT refreshed before the
                                                                   # test below (as in the reference, :1189-1200)
                if self.current_time + self.current_timestep > self.target_time + 1e-5:
                    self.current_timestep = self.target_time - self.current_time
                    self.override_timestep = True
            if self.current_time < self.target_time and self.override_timestep:    # :1213-1232
                self._call("force_timestep", "force_dt")(self.current_timestep)
                self.override_timestep = False
            if self.iterations_since_sync < self.rollback_limit and self.current_time < self.target_time:   # :1285-1304
                n = self.queue_addition_size
                self._call("step_batch", "run")(n)
                self.iterations_since_sync += n
                self.iterations += n
                self.cells_calculated += n * self.cells            # :1299: cols x rows per iteration, skipped or not
            self.read_key_statistics()                             # :1309-1313, blockUntilFinished, :1350
        finally:
            self.running = False


class Model:
    """CModel for one domain: `run()` is runModelMain."""

    def __init__(self, xml_path, make_sim=None, output_format=".npy", log=None, progress_interval=0.85,
                 clock=time.perf_counter):
        self.cfg = cfg = frontend.parse_configuration(xml_path)
        self.state0, self.bed, self.manning, self.res = frontend.build_domain(cfg)
        self.rows, self.cols = self.bed.shape
        if make_sim is None:
            from . import Domain
            sim = Domain(self.cols, self.rows, dx=self.res, scheme=cfg.scheme, precision=cfg.precision,
                         dry_threshold=cfg.dry_threshold, courant=cfg.courant, t_end=cfg.duration,
                         dynamic_dt=cfg.dynamic_dt, dt_fixed=cfg.timestep, dt_initial=cfg.timestep,
                         friction=cfg.friction, device=cfg.device_number - 1)
        else:
            sim = make_sim(cfg, self.cols, self.rows, self.res)
        self.sim = sim
        sim.upload(self.state0, self.bed, self.manning)
        frontend.attach_boundaries(cfg, sim, self.cols)
        self.scheme = SchemeDriver(sim, cfg, self.cols * self.rows)
        self.output_format, self.log, self.clock = output_format, log, clock
        self.progress_interval = progress_interval
        self.simulation_time = cfg.duration                        # dSimulationTime
        self.output_frequency = cfg.output_frequency
        self.current_time = self.target_time = self.last_sync_time = self.last_output_time = 0.0
        self.last_progress = 0.0
        self.outputs = []                                          # [(time, {value: array})]
        self.progress_blocks = []

    # CModel::runModelUpdateTarget (:723-770), one domain: run free until the next output is due
    def update_target(self):
        proposal = self.simulation_time
        if math.floor(proposal / self.output_frequency) > math.floor(self.last_sync_time / self.output_frequency):
            proposal = (math.floor(self.last_sync_time / self.output_frequency) + 1) * self.output_frequency
        self.target_time = proposal

    # CModel::runModelOutputs (:870-891) + CDomainCartesian::writeOutputs (:804-829)
    def write_outputs(self):
        due = abs(self.current_time - self.last_output_time - self.output_frequency) < 1e-5 and \
            self.current_time > self.last_output_time
        if not due:
            return False
        final = self.sim.download()
        out = {}
        for k, (what, pattern) in enumerate(self.cfg.targets):
            arr = frontend.derive_output(what, final, self.bed, self.res)
            out[what] = arr
            if pattern and self.cfg.target_dir and self.output_format:
                ext = self.output_format
                if ext == "xml":             # what the model file asks for: format="HFA" -> Imagine .img, anything else ESRI ASCII
                    fmt = (getattr(self.cfg, "target_formats", None) or [""] * (k + 1))[k]
                    ext = ".img" if fmt == "HFA" else ".asc"
                fname = os.path.splitext(pattern.replace("%t", str(int(round(self.current_time)))))[0] + ext
                frontend.write_raster(os.path.join(self.cfg.target_dir, fname), arr, self.res)
        self.outputs.append((self.current_time, out))
        self.last_output_time = self.current_time
        self.scheme.force_time_advance()
        if self.log:
            self.log(f"Output files written at {seconds_to_time(self.current_time)} ({len(out)} rasters)")
        return True

    # CModel::logProgress (:337-433): the same quantities; returned as a dict and, if a log sink is set, as a block
    def log_progress(self, seconds):
        t = min(self.current_time, self.simulation_time)
        progress = t / self.simulation_time if self.simulation_time > 0 else 1.0
        rate = int(self.scheme.cells_calculated / seconds) if seconds > 0 else 0
        remaining = min((1.0 - progress) * (seconds / progress), 31536000.0) if progress > 0 else 31536000.0
        block = dict(simulation_time=t, lowest_timestep=self.scheme.batch_timesteps, cells_calculated=self.scheme.cells_calculated,
                     rate=rate, processing_time=seconds, remaining=remaining, batch_size=self.scheme.queue_addition_size,
                     progress=progress)
        self.progress_blocks.append(block)
        if self.log:
            bar = "=" * max(0, int(math.floor(55 * progress)) - 1) + ">"
            self.log("\n".join([
                " SIMULATION PROGRESS",
                f" Simulation time:  {seconds_to_time(t):<15}Lowest timestep: {seconds_to_time(block['lowest_timestep']):>15}",
                f" Cells calculated: {block['cells_calculated']:<24}  Rate: {rate:>13}/s",
                f" Processing time:  {seconds_to_time(seconds):<16}Est. remaining: {seconds_to_time(remaining):>15}",
                f" Batch size:       {block['batch_size']:<16}",
                f" [{bar:<55}] {progress * 100:6.1f}%"]))
        return block

    # CModel::runModelMain (:1036-1100) for one local domain; the reference's polling of an asynchronous worker
    # collapses to a blocking batch per loop pass
    def run(self, max_outputs=None):
        t0 = self.clock()
        synchronised = True                                        # CModel.cpp:80: starts synchronised at t = 0
        while self.current_time < self.simulation_time - 1e-5:
            # runModelDomainAssess (:536-698)
            earliest = self.scheme.current_time
            sync_ready = self.scheme.is_sync_ready(self.target_time) and not synchronised and \
                self.last_sync_time != earliest
            synchronised = sync_ready
            self.current_time = earliest
            # runModelSync (:775-846): outputs, then a new target
            if synchronised or self.target_time == 0.0:
                self.last_sync_time = self.current_time
                self.write_outputs()
                if max_outputs is not None and len(self.outputs) >= max_outputs:
                    break
                self.update_target()
                synchronised = False
            # runModelSchedule (:905-957)
            self.scheme.run_simulation(self.target_time, self.clock() - t0)
            # runModelUI (:962-975)
            seconds = self.clock() - t0
            if seconds - self.last_progress > self.progress_interval:
                self.log_progress(seconds)
                self.last_progress = seconds
        # the loop ends once the last target is reached: one more assess + sync writes the final outputs
        self.current_time = self.scheme.current_time
        if self.scheme.is_sync_ready(self.target_time):
            self.write_outputs()
        self.seconds = self.clock() - t0
        self.log_progress(self.seconds)
        return self.outputs

    def close(self):
        if hasattr(self.sim, "close"):
            self.sim.close()
