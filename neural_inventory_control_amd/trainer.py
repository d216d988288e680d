"""`Trainer`: epoch loop, rollout, backward, optimizer step, dev evaluation, best-model tracking, early stopping and
checkpoints with the reference's signatures and file format (trainer.py:5-324).

`simulate_batch` has two routes with identical return values:
  * MLP policies the engine knows (`FusedRollout.supports`): the whole horizon forward AND backward runs in HIP kernels
    with no autograd graph; gradients land in `param.grad` and `do_one_epoch` skips `mean_loss.backward()`.
  * anything else (closed-form policies, user plugins): the reference's unrolled loop, with `Simulator.step` as one
    autograd-aware HIP kernel per period.
Multi-GPU (one process per GPU, scenarios sharded): `do_one_epoch` all-reduces [grads..., total, reported] once per
optimizer step through `parallel.GradientAllReducer` when torch.distributed is initialised.
"""
import copy
import datetime
import os

import numpy as np
import torch

from . import parallel
from .closed_form import ClosedFormRollout
from .gnn_rollout import GnnRollout
from .loss_functions import PolicyLoss
from .rollout import FusedRollout
from .tape_rollout import TapeRollout


def _numpy_scalar_globals():
    """What a reference-written checkpoint needs beyond tensors: numpy scalars / dtypes (best losses are np.float64)."""
    out = [np.dtype, np.float64, np.float32, np.int64, type(np.dtype("float64")), type(np.dtype("float32")),
           type(np.dtype("int64"))]
    for mod, name in (("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar")):
        try:
            out.append(getattr(__import__(mod, fromlist=[name]), name))
        except Exception:
            pass
    return out


class _FusedTotal(torch.autograd.Function):
    """Makes the fused rollout's total cost a differentiable function of the policy parameters for callers that use the
    reference idiom `total, _ = trainer.simulate_batch(...); (total / n).backward()` (trainer.py:160-173): the engine has
    already computed d(total)/d(theta) in its own backward sweep; autograd only scales and accumulates it."""

    @staticmethod
    def forward(ctx, total, n_params, *grads_and_params):
        ctx.save_for_backward(*grads_and_params[:n_params])
        return total.detach().clone()

    @staticmethod
    def backward(ctx, g):
        return (None, None) + (None,) * len(ctx.saved_tensors) + tuple(g * d for d in ctx.saved_tensors)


_OPTIMIZER_IMPL_KEYS = ("fused", "foreach", "capturable")


def _portable_optimizer_state(sd):
    """An optimizer state dict a DEFAULT torch.optim.Adam can load on any device - the reference's `load_model` calls
    `optimizer.load_state_dict` on one (trainer.py:300-312): implementation switches (`fused` & co.) are dropped from the parameter
    groups and the per-parameter `step` counters become host floats (a fused Adam keeps them as device tensors; loaded into a
    default Adam they would cost a device sync per step, and on a CPU-only box they cannot be stepped at all)."""
    out = {"state": {}, "param_groups": []}
    for k, st in sd.get("state", {}).items():
        st = dict(st)
        if torch.is_tensor(st.get("step")):
            st["step"] = torch.tensor(float(st["step"]), dtype=torch.float32)
        out["state"][k] = st
    for g in sd.get("param_groups", []):
        out["param_groups"].append({k: v for k, v in g.items() if k not in _OPTIMIZER_IMPL_KEYS})
    return out


class Trainer:
    def __init__(self, device="cpu"):
        self.all_train_losses = []
        self.all_dev_losses = []
        self.all_test_losses = []
        self.device = device
        self.time_stamp = self.get_time_stamp()
        self.best_performance_data = {"train_loss": np.inf, "dev_loss": np.inf, "last_epoch_saved": -1000,
                                      "model_params_to_save": None}
        self.best_epoch = 0
        self.use_fused_rollout = True
        self.use_tape_rollout = True   # quantile policies / just-in-time: batched decisions + one whole-horizon launch (tape_rollout.py)
        # generic route (policies the fused engine does not take: GNN, closed-form, user plugins): capture the whole training
        # step of a batch shape - every period's policy + env-step launches and the autograd sweep - into ONE HIP graph and
        # replay it; the per-period kernels of these policies run for microseconds, so the step is launch-bound
        # True / False / "auto" (default, round 4): the closed-form policies only - their whole-horizon kernel runs for 0.04 ms
        # inside ~0.5 ms of host work per step (level network, autograd, Adam bookkeeping), so their steps are always launch-bound
        # (base_stock 32,768 x T=100: 0.54 -> 0.17 ms); other generic-route policies keep eager steps unless asked
        self.use_step_graph = "auto"
        # MLP engine (per-period route): replay each rollout's launch sequence from a HIP graph - True / False / "auto" (the
        # engine measures host enqueue time against GPU time on a shape's second training step and replays only if the step is
        # launch-bound; `trainer_params.use_rollout_graph` in main_run)
        self.use_rollout_graph = "auto"
        # captured steps copy every batch tensor into the graph's static buffers before a replay; True = skip tensors presented
        # again unchanged (same object, address and version counter) - for callers that never rewrite a batch tensor in place
        # through a raw pointer (bench.py: one resident batch)
        self.inputs_versioned = False
        self._engines = {}
        self._step_graphs = {}
        self._fused_grads_ready = False

    def reset(self):
        self.all_train_losses, self.all_dev_losses, self.all_test_losses = [], [], []

    _hazard_guard = {}   # device index -> does this torch build REPORT the stream hazard `_graphed_generic_step` must see?

    @staticmethod
    def _autograd_stream_warnings(caught):
        """Warnings of a side-stream autograd run that mean "a gradient accumulator of this parameter is bound to another
        stream": recognised by WHERE they come from (torch's autograd package) or by the node they name - not by one sentence."""
        def hit(w):
            fn, msg = (getattr(w, "filename", "") or "").replace("\\", "/"), str(w.message)
            return "/torch/autograd/" in fn or "AccumulateGrad" in msg
        return [w for w in caught if hit(w)]

    @classmethod
    def _stream_hazard_is_reported(cls, device):
        """Self-test of the guard, once per device: build the very hazard on a 4-element parameter (an autograd graph alive on the
        default stream, a gradient taken on a side stream, nothing captured) and look for the report.  A torch build that stopped
        reporting it (reworded into another channel, or dropped) would leave `_graphed_generic_step` blind - and an invalidated
        capture ends the PROCESS in hipStreamEndCapture - so "auto" then never captures (explicit `use_step_graph = True` still
        does, at the caller's risk)."""
        import warnings
        dev = torch.device(device)
        if dev.type != "cuda":
            return False
        key = dev.index if dev.index is not None else torch.cuda.current_device()
        if key not in cls._hazard_guard:
            ok = False
            try:
                p_ = torch.nn.Parameter(torch.ones(4, device=dev))
                keep = (p_ * 2.0).sum()   # noqa: F841  (holds the graph, and with it the accumulator bound to the default stream)
                side = torch.cuda.Stream(device=dev)
                side.wait_stream(torch.cuda.current_stream(dev))
                was = torch.is_warn_always_enabled()
                torch.set_warn_always(True)
                try:
                    with warnings.catch_warnings(record=True) as caught:
                        warnings.simplefilter("always")
                        with torch.cuda.stream(side):
                            torch.autograd.grad((p_ * 3.0).sum(), [p_])
                finally:
                    torch.set_warn_always(was)
                torch.cuda.current_stream(dev).wait_stream(side)
                ok = bool(cls._autograd_stream_warnings(caught))
            except Exception:
                ok = False
            cls._hazard_guard[key] = ok
        return cls._hazard_guard[key]

    def _step_graph_on(self, model, observation_params):
        """`use_step_graph` resolved for a policy: "auto" = the closed-form policies (whole-horizon kernel inside the captured step)
        and the trainable quantile policies - provided this torch build reports the one condition under which a capture must
        be refused (`_stream_hazard_is_reported`)."""
        if self.use_step_graph == "auto":
            if not self.use_fused_rollout or not self._stream_hazard_is_reported(self.device):
                return False
            if ClosedFormRollout.supports(model) and self._plain_observation(observation_params):
                return True
            # the trainable quantile policies on the tape route: two whole-horizon launches inside ~40 small torch launches (the
            # batched forecaster pass, the interpolation and its autograd): launch-bound at every batch size the reference uses
            return bool(self.use_tape_rollout and TapeRollout.supports(model) and getattr(model, "trainable", True)
                        and type(model).__name__ != "JustInTime" and TapeRollout.observation_ok(model, observation_params))
        return bool(self.use_step_graph)

    # ---- training / evaluation loops (trainer.py:29-141) ---------------------------------------------------------
    def train(self, epochs, loss_function, simulator, model, data_loaders, optimizer, problem_params, observation_params,
              params_by_dataset, trainer_params):
        for epoch in range(epochs):
            _, train_report = self.do_one_epoch(
                optimizer, data_loaders["train"], loss_function, simulator, model, params_by_dataset["train"]["periods"],
                problem_params, observation_params, train=True, ignore_periods=params_by_dataset["train"]["ignore_periods"])
            self.all_train_losses.append(train_report)
            if epoch % trainer_params["do_dev_every_n_epochs"] == 0:
                _, dev_report = self.do_one_epoch(
                    optimizer, data_loaders["dev"], loss_function, simulator, model, params_by_dataset["dev"]["periods"],
                    problem_params, observation_params, train=False, ignore_periods=params_by_dataset["dev"]["ignore_periods"])
                self.all_dev_losses.append(dev_report)
                self.update_best_params_and_save(epoch, train_report, dev_report, trainer_params, model, optimizer)
                patience = trainer_params.get("early_stopping_patience_epochs", None)
                if patience is not None and (epoch - self.best_epoch) >= patience:
                    print(f"\nEarly stopping triggered at epoch {epoch + 1}")
                    print(f"No improvement for {epoch - self.best_epoch} epochs")
                    print(f"Best model was at epoch {self.best_epoch + 1} with dev loss: "
                          f"{self.best_performance_data['dev_loss']}")
                    break
            else:
                dev_report = 0
                self.all_dev_losses.append(self.all_dev_losses[-1])
            if epoch % trainer_params["print_results_every_n_epochs"] == 0 and parallel.rank() == 0:
                print(f"epoch: {epoch + 1}")
                print(f"Average per-period train loss: {train_report}")
                print(f"Average per-period dev loss: {dev_report}")
                print(f"Best per-period dev loss: {self.best_performance_data['dev_loss']}")

    def test(self, loss_function, simulator, model, data_loaders, optimizer, problem_params, observation_params,
             params_by_dataset, discrete_allocation=False):
        if model.trainable and self.best_performance_data["model_params_to_save"] is not None:
            model.load_state_dict(self.best_performance_data["model_params_to_save"])
        return self.do_one_epoch(
            optimizer, data_loaders["test"], loss_function, simulator, model, params_by_dataset["test"]["periods"],
            problem_params, observation_params, train=False, ignore_periods=params_by_dataset["test"]["ignore_periods"],
            discrete_allocation=discrete_allocation)

    def do_one_epoch(self, optimizer, data_loader, loss_function, simulator, model, periods, problem_params,
                     observation_params, train=True, ignore_periods=0, discrete_allocation=False):
        """trainer.py:143-179.  Loss scalars are accumulated on the device and read back ONCE per epoch (the reference
        calls .item() twice per batch, :166-167)."""
        epoch_loss = torch.zeros((), device=self.device, dtype=torch.float64)
        epoch_report = torch.zeros((), device=self.device, dtype=torch.float64)
        total_samples = len(data_loader.dataset)
        tracked = periods - ignore_periods
        world = parallel.world_size()
        sharded = parallel.active()  # a process group exists (also with one rank): run the step's collective
        with torch.no_grad() if not train else torch.enable_grad():
            for data_batch in data_loader:
                data_batch = self.move_batch_to_device(data_batch)
                step_graph = self._step_graph_on(model, observation_params)
                if train and not step_graph:
                    optimizer.zero_grad()
                elif train:
                    optimizer.zero_grad(set_to_none=False)  # captured steps accumulate into fixed .grad tensors
                self._fused_grads_ready = False
                train_now = bool(train and model.trainable)
                # every rank normalises by the GLOBAL batch (trainer.py:169 with B summed over ranks)
                global_batch = getattr(data_loader, "last_global_batch", None) or len(data_batch["demands"]) * world
                if len(data_batch["demands"]) == 0:
                    # a sharded job whose last global batch is smaller than the world size leaves this rank without
                    # scenarios: it contributes zeros but still joins the step's collective and steps the optimizer
                    # (and adds the step's GLOBAL totals, so that every rank reports - and early-stops on - the same losses)
                    zero = torch.zeros((), device=self.device)
                    tot, rep = zero, zero
                    if train and model.trainable and sharded:
                        tot, rep = parallel.GradientAllReducer.get(model).all_reduce(zero, zero)
                        if getattr(model, "gradient_clipping_norm_value", None) is not None:
                            torch.nn.utils.clip_grad_norm_(model.parameters(), model.gradient_clipping_norm_value)
                        optimizer.step()
                    elif sharded:
                        tot, rep = parallel.all_reduce_scalars(zero, zero)
                    epoch_loss += tot.detach()
                    epoch_report += rep.detach()
                    continue
                # `use_step_graph` captures the whole training step of every policy whose per-period work is launch-bound: the
                # generic route AND the closed-form policies' whole-horizon kernel (its launch, the level network's autograd
                # and the totals: 0.55 -> 0.18 ms per step).  The MLP / GNN engines run GPU-bound launch sequences of their own
                # (`FusedRollout.use_graph` replays those) and keep the eager step.
                graphed = (train and model.trainable and step_graph and not discrete_allocation
                           and not (self.use_fused_rollout
                                    and ((FusedRollout.supports(model) and FusedRollout.observation_ok(model, observation_params))
                                         or (self._plain_observation(observation_params)
                                             and GnnRollout.supports(model, problem_params)))))
                if graphed:
                    total_reward, reward_to_report = self._graphed_generic_step(
                        loss_function, simulator, model, periods, problem_params, data_batch, observation_params,
                        ignore_periods, global_batch=global_batch)
                else:
                    total_reward, reward_to_report = self.simulate_batch(
                        loss_function, simulator, model, periods, problem_params, data_batch, observation_params,
                        ignore_periods, discrete_allocation, train=train_now, global_batch=global_batch)
                if train and model.trainable:
                    if not self._fused_grads_ready:
                        mean_loss = total_reward / (global_batch * periods * problem_params["n_stores"])
                        mean_loss.backward()
                    if sharded:
                        total_reward, reward_to_report = parallel.GradientAllReducer.get(model).all_reduce(
                            total_reward.detach(), reward_to_report.detach())
                    if getattr(model, "gradient_clipping_norm_value", None) is not None:
                        torch.nn.utils.clip_grad_norm_(model.parameters(), model.gradient_clipping_norm_value)
                    optimizer.step()
                elif sharded:
                    total_reward, reward_to_report = parallel.all_reduce_scalars(total_reward.detach(),
                                                                                 reward_to_report.detach())
                epoch_loss += total_reward.detach()
                epoch_report += reward_to_report.detach()
        # in a sharded job `data_loader.dataset` is the GLOBAL dataset (each rank iterates its slice of every batch)
        n_stores = problem_params["n_stores"]
        return (epoch_loss.item() / (total_samples * periods * n_stores),
                epoch_report.item() / (total_samples * tracked * n_stores))

    def simulate_batch(self, loss_function, simulator, model, periods, problem_params, data_batch, observation_params,
                       ignore_periods=0, discrete_allocation=False, *, train=None, global_batch=None):
        """trainer.py:181-216.  `train` / `global_batch` are passed by `do_one_epoch` (the engine then writes
        d(mean_loss)/d(theta) straight into `param.grad`).  A DIRECT call (the reference's public API, e.g. from a notebook)
        leaves them None: gradients are wanted iff autograd is recording and the policy is trainable, and the returned total
        is a differentiable function of the parameters, so `(total / n).backward()` works as it does upstream."""
        direct = train is None
        if direct:
            train = torch.is_grad_enabled() and bool(getattr(model, "trainable", True))
        train = bool(train) and torch.is_grad_enabled()
        # (rounded actions have zero gradient: a training step with discrete allocation keeps the reference's generic route;
        # a custom loss module is honoured by the generic route - the fused engines implement PolicyLoss = reward.sum())
        engine_ok = self.use_fused_rollout and not (discrete_allocation and train) and isinstance(loss_function, PolicyLoss)
        fusable = engine_ok and self._plain_observation(observation_params)
        if fusable and ClosedFormRollout.supports(model):
            # closed-form policies: ONE kernel for the whole horizon, gradient included (forward mode); the returned total
            # is an ordinary differentiable tensor, so the caller's mean_loss.backward() reaches the policy's parameters
            # (a captured step holds the engine's buffers by address: with `use_step_graph` every batch shape keeps its own
            # engine, otherwise an epoch's smaller last batch would re-size - free - what the first graph replays into)
            # ... and every (horizon, training / evaluation) context: a dev pass with the training batch size but another
            # `periods` would otherwise re-size the buffers the captured training step replays into by raw address
            ekey = ((id(model), "closed_form", len(data_batch["demands"]), periods, bool(train))
                    if self._step_graph_on(model, observation_params) else (id(model), "closed_form", None))
            eng = self._engines.get(ekey)
            if eng is None or eng.model is not model:
                eng = self._engines[ekey] = ClosedFormRollout(model, problem_params, self.device)
            if eng.shapes_ok(data_batch):
                self._last_engine = eng
                return eng.run(data_batch, periods, ignore_periods, train=train, observation_params=observation_params,
                               discrete_allocation=discrete_allocation)
        if engine_ok and self.use_tape_rollout and TapeRollout.supports(model) \
                and TapeRollout.observation_ok(model, observation_params, data_batch):
            # quantile policies / just-in-time: the decisions of all periods do not depend on the state - one batched pass computes
            # them (the forecaster runs once), one whole-horizon launch per direction does the rest (tape_rollout.py); the
            # returned total is differentiable with respect to the policy's parameters like the closed-form engine's
            # (a captured step holds the engine's buffers by address: with a step graph every (batch size, horizon, train / eval)
            # context keeps its own engine - as for the closed-form engine above)
            ekey = ((id(model), "tape", len(data_batch["demands"]), periods, bool(train))
                    if self._step_graph_on(model, observation_params) else (id(model), "tape", bool(train)))
            eng = self._engines.get(ekey)
            if eng is None or eng.model is not model:
                eng = self._engines[ekey] = TapeRollout(model, problem_params, self.device)
            if eng.shapes_ok(data_batch, periods, (observation_params["demand"] or {}).get("period_shift") or 0):
                self._last_engine = eng
                return eng.run(data_batch, periods, ignore_periods, train=train, observation_params=observation_params,
                               discrete_allocation=discrete_allocation)
        engine_cls = None
        # (the MLP engine builds the observation itself: inventories for the vanilla policies, + the past-demand window and the
        # days-from-christmas feature for data_driven; everything else with a moving observation keeps the generic loop)
        mlp_ok = engine_ok and FusedRollout.supports(model) and FusedRollout.observation_ok(model, observation_params, data_batch)
        if mlp_ok:
            engine_cls = FusedRollout    # (also evaluation with discrete allocation: the MLP engine rounds in-kernel)
        elif fusable and not discrete_allocation:
            if GnnRollout.supports(model, problem_params) and "mean" in data_batch and "std" in data_batch:
                engine_cls = GnnRollout  # fused gather-MLP kernels over the static supply graph (gnn_rollout.py)
        if engine_cls is not None:
            # one engine per (policy, training / evaluation): an epoch alternates a training pass and a dev pass with
            # different horizons and buffer needs, and re-sizing one engine back and forth would reallocate tens of GB
            eng = self._engines.get((id(model), train))
            if eng is None or eng.model is not model:  # (the engine holds the model, so its id cannot be recycled)
                eng = self._engines[(id(model), train)] = engine_cls(model, problem_params, self.device)
            # A batch of ANOTHER shape (the short last batch of an epoch) gets its own engine while a graph mode is on: re-sizing
            # one engine back and forth drops its buffers, its captured graphs and its measured replay decision - every epoch
            # would pay a reallocation, an eager run, a probed run and a fresh capture for each of the two shapes.
            shape = (len(data_batch["demands"]), periods)
            first = getattr(eng, "_trainer_shape", None)
            if first is None:
                eng._trainer_shape = shape
            elif first != shape and self.use_rollout_graph in (True, "auto"):
                key2 = (id(model), shape, train)
                eng2 = self._engines.get(key2)
                if eng2 is None or eng2.model is not model:
                    others = [k for k in self._engines if len(k) == 3 and k[0] == id(model) and isinstance(k[1], tuple) and k[2] == train]
                    for k in others[:-1]:   # (at most two extra shapes per policy and pass: the oldest goes)
                        del self._engines[k]
                    eng2 = self._engines[key2] = engine_cls(model, problem_params, self.device)
                    eng2._trainer_shape = shape
                eng = eng2
            eng.use_graph = self.use_rollout_graph   # (both engines: True / False / "auto" = by measurement per shape)
            if hasattr(eng, "zero_lead_orders"):   # (the Simulator's rule for orders without a lead time also holds on the fused route)
                eng.zero_lead_orders = getattr(simulator, "zero_lead_orders", "drop")
            for opt_ in ("fuse_tail", "use_wide", "use_period_kernel", "use_period_bwd"):   # (A/B switches of the engines' routes: set on the trainer, handed on)
                if hasattr(self, opt_) and hasattr(eng, opt_):
                    setattr(eng, opt_, getattr(self, opt_))
            if direct and train:
                total, reported = eng.run(data_batch, periods, ignore_periods, train=True,
                                          observation_params=observation_params, grad_scale=1.0, assign_grads=False)
                pg = eng.param_grads()
                total = _FusedTotal.apply(total, len(pg), *[g.clone() for _, g in pg], *[p for p, _ in pg])
                return total, reported
            gb = global_batch if global_batch is not None else len(data_batch["demands"])
            total, reported = eng.run(data_batch, periods, ignore_periods, train=train,
                                      observation_params=observation_params,
                                      grad_scale=1.0 / (gb * periods * problem_params["n_stores"]),
                                      discrete_allocation=discrete_allocation)
            self._fused_grads_ready = train
            return total, reported

        self._last_engine = None
        batch_reward, reward_to_report = 0, 0
        observation, _ = simulator.reset(periods, problem_params, data_batch, observation_params)
        for t in range(periods):
            obs_and_internal = {k: v for k, v in observation.items()}
            obs_and_internal["internal_data"] = simulator._internal_data
            action = model(obs_and_internal)
            if discrete_allocation:
                action = {k: v.round() for k, v in action.items()}
            observation, reward, terminated, _, _ = simulator.step(action)
            total_reward = loss_function(None, action, reward)
            batch_reward += total_reward
            if t >= ignore_periods:
                reward_to_report += total_reward
            if terminated:
                break
        return batch_reward, reward_to_report

    def _graphed_generic_step(self, loss_function, simulator, model, periods, problem_params, data_batch,
                              observation_params, ignore_periods, global_batch=None):
        """Forward + backward of one batch on the generic route, replayed from a HIP graph.  First call with a batch shape:
        eager (materialises lazy layers, compiles static policy state); second call: captured with the batch copied into
        static input tensors and gradients accumulating into static `.grad` tensors; afterwards: copy + replay."""
        key = (id(model), periods, ignore_periods,
               tuple((k, tuple(v.shape)) for k, v in sorted(data_batch.items()) if torch.is_tensor(v)))
        st = self._step_graphs.get(key)
        if st is not None and st["model"] is not model:  # id(model) recycled after a garbage collection
            st = None
        if global_batch is None:
            global_batch = getattr(self, "_global_batch", None) or len(data_batch["demands"])
        scale = 1.0 / (global_batch * periods * problem_params["n_stores"])
        if st is None or st.get("eager_only"):  # eager warm-up run (or a shape whose capture was refused, see below)
            if st is None:
                self._step_graphs[key] = {"static": None, "model": model}
            total, rep = self.simulate_batch(loss_function, simulator, model, periods, problem_params, data_batch,
                                             observation_params, ignore_periods, False, train=True)
            (total * scale).backward()
            self._fused_grads_ready = True
            return total.detach(), rep.detach() if torch.is_tensor(rep) else rep
        params = [p for p in model.parameters() if p.requires_grad]
        if st["static"] is None:
            static = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in data_batch.items()}
            # static gradient tensors: the captured graph writes into THESE.  They are shared by every captured batch shape
            # of the model (an epoch that ends on a smaller batch captures a second graph: replacing `.grad` here would
            # leave the first graph accumulating into tensors the optimizer no longer sees)
            for p in params:
                if p.grad is None:
                    p.grad = torch.zeros_like(p)
            st["grads"] = [p.grad for p in params]
            stream = torch.cuda.Stream()
            stream.wait_stream(torch.cuda.current_stream())
            # Gradients go into the static tensors through torch.autograd.grad + copy_, NOT through .backward(): a parameter's
            # AccumulateGrad node is bound to the stream of the graph that created it and lives as long as any earlier result
            # the caller still holds (a `total` of an eager simulate_batch on the default stream, say) - re-used from the capture
            # stream it invalidates the capture, and ending an invalidated capture takes the process down on this stack.
            def grads_into_static(loss):
                for p, g_ in zip(params, torch.autograd.grad(loss, params, allow_unused=True)):
                    if g_ is None:
                        p.grad.zero_()
                    else:
                        p.grad.copy_(g_)
            import warnings
            warn_always = torch.is_warn_always_enabled()
            torch.set_warn_always(True)   # (the hazard below is reported through a warn-once channel)
            try:
                with warnings.catch_warnings(record=True) as caught:
                    warnings.simplefilter("always")
                    with torch.cuda.stream(stream):  # one more eager run on the side stream (allocator / autograd warm-up)
                        total, rep = self.simulate_batch(loss_function, simulator, model, periods, problem_params, static,
                                                         observation_params, ignore_periods, False, train=True)
                        grads_into_static(total * scale)
            finally:
                torch.set_warn_always(warn_always)
            torch.cuda.current_stream().wait_stream(stream)
            pinned = self._autograd_stream_warnings(caught)
            for w in caught:
                if w not in pinned:
                    warnings.warn_explicit(w.message, w.category, w.filename, w.lineno)
            if pinned:
                # Something outside still holds an autograd graph of this policy from an eager step on the default stream (a
                # `total` returned by simulate_batch, say): the parameters' gradient accumulators stay bound to that stream, the
                # autograd engine then synchronises the capture stream with it, and that invalidates a capture (on this stack the
                # process dies in hipStreamEndCapture).  This batch shape keeps eager steps; the run above already was one.
                warnings.warn("use_step_graph: an autograd graph from an earlier eager step of this policy is still alive (e.g. a "
                              "returned `total` that was not released); its training steps are not captured into a HIP graph")
                st["eager_only"] = True
                self._fused_grads_ready = True
                return total.detach(), rep.detach() if torch.is_tensor(rep) else rep
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                total, rep = self.simulate_batch(loss_function, simulator, model, periods, problem_params, static,
                                                 observation_params, ignore_periods, False, train=True)
                grads_into_static(total * scale)
                out_total, out_rep = total.detach(), (rep.detach() if torch.is_tensor(rep) else rep)
            # the captured env-step launches read the compact cost / lead-time tables of THIS EnvProblem (built from
            # `static` at capture time), not `static` itself: they are refreshed from every incoming batch below
            # (a closed-form policy's whole-horizon engine owns the EnvProblem its captured launch reads)
            eng_ = getattr(self, "_last_engine", None)
            st.update(static=static, graph=graph, total=out_total, rep=out_rep, scale=scale,
                      prob=simulator._prob if eng_ is None else eng_.prob)
        if st["scale"] != scale:
            raise RuntimeError("captured training step: the global batch size changed; disable use_step_graph")
        incoming = simulator._problem_for(problem_params, {k: v for k, v in data_batch.items() if torch.is_tensor(v)},
                                          torch.device(self.device))
        if incoming is not st["prob"]:
            if not st["prob"].same_layout(incoming):
                # e.g. the captured batch had scenario-uniform tables and this one varies across samples: re-capture
                del self._step_graphs[key]
                return self._graphed_generic_step(loss_function, simulator, model, periods, problem_params, data_batch,
                                                  observation_params, ignore_periods, global_batch=global_batch)
            st["prob"].copy_tables_from(incoming)
        seen = st.setdefault("seen", {})
        for k, v in data_batch.items():
            if torch.is_tensor(v):
                # (opt-in `inputs_versioned`, as on the fused engines: a tensor presented again with the same identity, address and
                # version counter is not copied a second time - at 10^6 chains the demand trace alone is 419 MB per step; a tensor
                # rewritten behind torch's back, without a version bump, would be missed, hence not the default)
                tag = (id(v), v.data_ptr(), v._version)
                if self.inputs_versioned and seen.get(k) == tag:
                    continue
                st["static"][k].copy_(v)
                seen[k] = tag
        for p, g in zip(params, st["grads"]):  # (someone set `.grad` to None / another tensor since the capture)
            if p.grad is not g:
                p.grad = g
        st["graph"].replay()
        self._fused_grads_ready = True
        return st["total"], st["rep"]

    @staticmethod
    def _plain_observation(observation_params):
        return (observation_params["demand"]["past_periods"] == 0 and not observation_params["time_features"]
                and not observation_params["sample_features"])

    # ---- checkpoints (trainer.py:218-276, 300-312): same dict keys, including the reference's mislabeled ones -----
    def save_model(self, epoch, model, optimizer, trainer_params):
        path = self.create_many_folders_if_not_exist_and_return_path(trainer_params["base_dir"],
                                                                     trainer_params["save_model_folders"])
        torch.save({
            "epoch": epoch,
            "model_state_dict": self.best_performance_data["model_params_to_save"],
            "optimizer_state_dict": _portable_optimizer_state(optimizer.state_dict()),
            "best_train_loss": self.best_performance_data["dev_loss"],
            "best_dev_loss": self.all_train_losses,
            "all_train_losses": self.all_train_losses,
            "all_dev_losses": self.all_dev_losses,
            "all_test_losses": self.all_test_losses,
            "warehouse_upper_bound": model.warehouse_upper_bound,
        }, f"{path}/{trainer_params['save_model_filename']}.pt")

    def create_folder_if_not_exists(self, folder):
        if not os.path.isdir(folder):
            os.mkdir(folder)

    def create_many_folders_if_not_exist_and_return_path(self, base_dir, intermediate_folder_strings):
        path = base_dir
        for s in intermediate_folder_strings:
            path += f"/{s}"
            self.create_folder_if_not_exists(path)
        return path

    def update_best_params_and_save(self, epoch, train_loss, dev_loss, trainer_params, model, optimizer):
        current = {"train_loss": train_loss, "dev_loss": dev_loss}
        key = trainer_params["choose_best_model_on"]
        if current[key] < self.best_performance_data[key]:
            self.best_performance_data["train_loss"] = train_loss
            self.best_performance_data["dev_loss"] = dev_loss
            if model.trainable:
                self.best_performance_data["model_params_to_save"] = copy.deepcopy(model.state_dict())
            self.best_performance_data["update"] = True
            self.best_epoch = epoch
        if trainer_params["save_model"] and model.trainable and parallel.rank() == 0:
            if self.best_performance_data["last_epoch_saved"] + trainer_params["epochs_between_save"] <= epoch \
                    and self.best_performance_data.get("update"):
                self.best_performance_data["last_epoch_saved"] = epoch
                self.best_performance_data["update"] = False
                self.save_model(epoch, model, optimizer, trainer_params)

    def plot_losses(self, ymin=None, ymax=None):
        import matplotlib.pyplot as plt
        plt.plot(self.all_train_losses, label="Train loss")
        plt.plot(self.all_dev_losses, label="Dev loss")
        plt.legend()
        if ymin is not None and ymax is not None:
            plt.ylim(ymin, ymax)
        plt.xlabel("Epoch")
        plt.ylabel("Loss")
        plt.show()

    def move_batch_to_device(self, data_batch):
        return {k: v.to(self.device) for k, v in data_batch.items()}

    def load_model(self, model, optimizer, model_path):
        # The reference's checkpoints hold tensors, lists and numbers (numpy scalars among them: `best_train_loss` is a
        # np.float64): they are allow-listed for the safe loader instead of unpickling arbitrary objects.  A checkpoint that
        # still fails the safe load is only unpickled when the caller opted in (`trainer.allow_pickled_checkpoints = True`).
        import pickle
        try:
            with torch.serialization.safe_globals(_numpy_scalar_globals()):
                checkpoint = torch.load(model_path, map_location=self.device, weights_only=True)
        except pickle.UnpicklingError as e:
            if not getattr(self, "allow_pickled_checkpoints", False):
                raise pickle.UnpicklingError(
                    f"{model_path}: not loadable with weights_only=True ({e}); set trainer.allow_pickled_checkpoints = True "
                    "to unpickle it (only for files you trust)") from e
            import warnings
            warnings.warn(f"{model_path}: falling back to the unsafe pickle loader (allow_pickled_checkpoints)")
            checkpoint = torch.load(model_path, map_location=self.device, weights_only=False)
        model.load_state_dict(checkpoint["model_state_dict"])
        # (the implementation switches of THIS optimizer - fused / foreach / capturable - stay its own: a checkpoint written by a
        # fused Adam must not turn a default one into a fused one, nor the other way round)
        own = [{k: g.get(k) for k in _OPTIMIZER_IMPL_KEYS if k in g} for g in optimizer.param_groups]
        portable = _portable_optimizer_state(checkpoint["optimizer_state_dict"])
        # torch decides where each `step` counter lives from the SAVED groups' fused / capturable flags: hand it this optimizer's
        # own, so that a fused Adam gets its counters as device tensors and a default one keeps host floats
        for g, keep in zip(portable["param_groups"], own):
            g.update(keep)
        optimizer.load_state_dict(portable)
        for g, keep in zip(optimizer.param_groups, own):
            g.update(keep)
        for g in optimizer.param_groups:
            on_device = bool(g.get("fused")) or bool(g.get("capturable"))
            for p in g["params"]:
                step = optimizer.state.get(p, {}).get("step")
                if torch.is_tensor(step) and on_device and step.device != p.device:
                    optimizer.state[p]["step"] = step.to(device=p.device, dtype=torch.float32)
        self.all_train_losses = checkpoint["all_train_losses"]
        self.all_dev_losses = checkpoint["all_dev_losses"]
        self.all_test_losses = checkpoint["all_test_losses"]
        model.warehouse_upper_bound = checkpoint["warehouse_upper_bound"]
        return model, optimizer

    def get_time_stamp(self):
        return int(datetime.datetime.now().timestamp())

    def get_year_month_day(self):
        ct = datetime.datetime.now()
        return f"{ct.year}_{ct.month:02d}_{ct.day:02d}"
