"""Host-side mirror of src/Network/TrainerController.py: same constructor, same methods
(init_model_dir, restore_model, train_network, train_step, test_step, save_best_model, quicksave), same loss.csv
columns; the arithmetic runs in lib4dflow_hip.so.

Differences that are deliberate and documented in DESIGN.md:
  * the TensorBoard epoch scalars (:181-182, :396-412) are written by tfevents.SummaryWriter -- same directories, tags,
    steps and TF2 scalar encoding, no TensorFlow dependency;
  * metrics accumulate on the device and are only synchronised when .result() is read, so a training loop that
    does not print every step never stalls the GPU (the reference forces a sync per step, :290);
  * with torch.distributed initialised, train_step sum-all-reduces the flat gradient over RCCL before Adam."""
import datetime
import os
import pickle
import shutil
import time

import numpy as np
import torch

from . import ops, parallel, tfevents
from .network import Input, SR4DFlowNet

L2_LAMBDA = 5e-7                      # SR4DFlowNet.py:99
ADAM_B1, ADAM_B2, ADAM_EPS = 0.9, 0.999, 1e-7   # tf.keras.optimizers.Adam defaults (TrainerController.py:73)


class Mean:
    """tf.keras.metrics.Mean: running mean over every element ever passed to update_state."""

    def __init__(self, name, device):
        self.name = name
        self.device = device
        self.reset_states()

    def reset_states(self):
        self._total = torch.zeros((), device=self.device, dtype=torch.float64)
        self._count = 0

    def update_state(self, values):
        if isinstance(values, torch.Tensor):
            self._total += values.sum().to(torch.float64)
            self._count += values.numel()
        else:
            self._total += float(values)
            self._count += 1

    def result(self):
        return float(self._total.item()) / self._count if self._count else 0.0

    def result_global(self):
        """Mean over every rank's elements (collective: every rank must call it).  Single process == result()."""
        total, count = parallel.allreduce_sum_host([float(self._total.item()), float(self._count)])
        return total / count if count else 0.0


class _Optimizer:
    """Keras-Adam state on one flat buffer.  `weights` mirrors optimizer.weights = [iterations, m..., v...]."""

    def __init__(self, model, lr):
        self.lr = float(lr)
        self.iterations = 0
        self.m = torch.zeros_like(model.flat_w)
        self.v = torch.zeros_like(model.flat_w)

    def lr_t(self):
        t = self.iterations
        return self.lr * np.sqrt(1.0 - ADAM_B2 ** t) / (1.0 - ADAM_B1 ** t)


class TrainerController:
    def __init__(self, patch_size, res_increase, initial_learning_rate=1e-4, quicksave_enable=True,
                 network_name='4DFlowNet', low_resblock=8, hi_resblock=4, device=None, seed=0, dtype='float32',
                 bucketed_allreduce=None, conv_algo=None):
        """Reference arguments: TrainerController.py:18.  Extra (keyword-only in spirit): device, seed (Glorot draw), dtype
        (activation storage, 'float32' | 'bfloat16'), bucketed_allreduce (data parallel only: True = one asynchronous SUM
        all-reduce per gradient bucket started inside backward -- the default --, False = ONE all-reduce of the whole buffer
        after backward; env FDN_DP_BUCKETED=0 selects the latter when the argument is None), conv_algo ('auto' | 'direct' |
        {layer name: ...}: algorithm of the 64->64 layers, see FlowNetModel)."""
        self.div_weight = 0            # divergence loss is dead code in the reference (TrainerController.py:23,121)
        self.non_fluid_weight = 1
        self.res_increase = res_increase
        self.patch_size = patch_size
        self.QUICKSAVE_ENABLED = quicksave_enable
        self.network_name = network_name

        input_shape = (patch_size, patch_size, patch_size, 1)
        u, v, w = Input(input_shape, 'u'), Input(input_shape, 'v'), Input(input_shape, 'w')
        u_mag, v_mag, w_mag = Input(input_shape, 'u_mag'), Input(input_shape, 'v_mag'), Input(input_shape, 'w_mag')
        net = SR4DFlowNet(res_increase)
        self.model = net.build_network(u, v, w, u_mag, v_mag, w_mag, low_resblock, hi_resblock, device=device, seed=seed,
                                       dtype=dtype, conv_algo=conv_algo)
        self.device = self.model.device

        names = ['train_loss', 'val_loss', 'train_accuracy', 'val_accuracy', 'train_mse', 'val_mse', 'train_div',
                 'val_div', 'l2_reg_loss']
        self.loss_metrics = dict((n, Mean(n, self.device)) for n in names)
        self.accuracy_metric = 'val_loss'
        self.learning_rate = initial_learning_rate
        self.optimizer = _Optimizer(self.model, initial_learning_rate)
        self._l2_buf = torch.zeros(1, device=self.device)
        self._l2_partials = torch.zeros(ops.ADAM_PARTIALS, device=self.device)   # sum(w^2) blocks left by the last Adam step
        self._l2_version = -1          # model.weights_version those partials belong to
        self.unique_model_name = network_name
        self.model_dir = None
        if bucketed_allreduce is None:
            bucketed_allreduce = os.environ.get("FDN_DP_BUCKETED", "1") not in ("0", "false", "no")
        self.bucketed_allreduce = bool(bucketed_allreduce)
        # bench / diagnosis: with profile_allreduce set, every train_step appends a pair of HIP events that bracket the point where
        # the compute stream waits for the gradient all-reduce(s): their distance is the EXPOSED collective time of that step
        self.profile_allreduce = False
        self.allreduce_wait_events = []

    # ------------------------------------------------------------------ steps
    def _unpack(self, data_pairs):
        arrs = [self.model._to_dev(a) for a in data_pairs]
        u, v, w, u_mag, v_mag, w_mag, u_hr, v_hr, w_hr, venc, mask = arrs
        return (u, v, w, u_mag, v_mag, w_mag), (u_hr, v_hr, w_hr), venc, mask

    def calculate_regularizer_loss(self):
        """5e-7 * sum(kernel^2) as a 0-d device tensor (TrainerController.py:129-141)."""
        if self._l2_version != self.model.weights_version:         # first step, or weights were loaded / re-initialised: stream them once
            ops.l2_sumsq_partials(self.model.flat_w, self.model.is_kernel, self._l2_partials)
            self._l2_version = self.model.weights_version
        ops.sum_partials(self._l2_partials, self._l2_buf)          # (else the Adam kernel left the per-block sums behind)
        return self._l2_buf[0] * L2_LAMBDA

    def calculate_and_update_metrics(self, hires, predictions, mask, metric_set, want_grad):
        out, dpred = ops.loss_metrics(predictions, hires[0], hires[1], hires[2], mask, want_grad=want_grad)
        mse, rel_error = out[:, 0], out[:, 1]
        loss = mse
        if metric_set == 'train':
            l2 = self.calculate_regularizer_loss()
            self.loss_metrics['l2_reg_loss'].update_state(l2.reshape(1))
            loss = mse + l2
        self.loss_metrics['%s_loss' % metric_set].update_state(loss)
        self.loss_metrics['%s_mse' % metric_set].update_state(mse)
        self.loss_metrics['%s_div' % metric_set].update_state(0.0)
        self.loss_metrics['%s_accuracy' % metric_set].update_state(rel_error)
        return loss, dpred

    def train_step(self, data_pairs):
        """TrainerController.py:209-225: forward, loss (+L2), gradient of sum_b loss_b, Adam."""
        inputs, hires, venc, mask = self._unpack(data_pairs)
        B = inputs[0].shape[0]
        m = self.model
        # The trailing slot of the gradient buffer carries this rank's batch size; after the SUM all-reduce it holds the
        # global batch size, which the Adam kernel reads on the device -- no host synchronisation per step.
        m.batch_slot.fill_(float(B))
        # Data parallel: one SUM all-reduce per gradient bucket, started as soon as backward() has enqueued the bucket's last
        # launch (RCCL works on its own stream: the hi-res bucket travels while the low-res layers are still computing).  Every
        # rank issues the same buckets in the same order, also a rank whose shard of a ragged batch is empty.
        pending = []
        reduce_bucket = None
        dp = parallel.world_size() > 1
        if dp and self.bucketed_allreduce:
            def reduce_bucket(lo, hi):
                pending.append(parallel.allreduce_sum_start(m.flat_g_ext[lo:hi]))
        if B > 0:
            pred = m.forward(inputs, training=True)
            loss, dpred = self.calculate_and_update_metrics(hires, pred, mask, 'train', True)
            m.backward(dpred, grad_ready=reduce_bucket)
        else:                                   # ragged tail on this rank: contribute a zero gradient
            m.flat_g.zero_()
            loss = None
            if reduce_bucket is not None:
                for lo, hi in m.grad_buckets:
                    reduce_bucket(lo, hi)
        if dp and not self.bucketed_allreduce:             # the plain form: one collective over the whole buffer after backward
            pending.append(parallel.allreduce_sum_start(m.flat_g_ext))
        if dp and self.profile_allreduce:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        for h in pending:
            parallel.allreduce_wait(h)
        if dp and self.profile_allreduce:
            e1.record()
            self.allreduce_wait_events.append((e0, e1))
            del self.allreduce_wait_events[:-4096]          # a diagnostic left on outside the bench must not grow without bound
        opt = self.optimizer
        opt.iterations += 1
        # L2 regulariser gradient: the (B,) loss vector carries the scalar L2 term B times (:249) -> B_global * 2*lambda*w
        ops.adam_step(m.flat_w, m.flat_g, opt.m, opt.v, m.is_kernel, opt.lr_t(), ADAM_B1, ADAM_B2, ADAM_EPS,
                      2.0 * L2_LAMBDA, m.batch_slot, sumsq_partials=self._l2_partials)
        m.weights_changed()
        self._l2_version = m.weights_version
        return loss

    def test_step(self, data_pairs):
        """TrainerController.py:227-239: forward + metrics, no L2, no update."""
        inputs, hires, venc, mask = self._unpack(data_pairs)
        if inputs[0].shape[0] == 0:             # ragged tail on this rank (data parallel): nothing to evaluate
            return None
        pred = self.model.forward(inputs, training=False)
        self.calculate_and_update_metrics(hires, pred, mask, 'val', False)
        return pred

    def reset_metrics(self):
        for k in self.loss_metrics:
            self.loss_metrics[k].reset_states()

    # ------------------------------------------------------------------ directories / logging
    def init_model_dir(self, base_dir="../models"):
        timestamp = datetime.datetime.now().strftime("%Y%m%d-%H%M")
        self.unique_model_name = '%s_%s' % (self.network_name, timestamp)
        self.model_dir = "%s/%s" % (base_dir, self.unique_model_name)
        self.model_path = "%s/%s" % (self.model_dir, self.network_name)
        if parallel.rank() == 0:
            os.makedirs(self.model_dir, exist_ok=True)
            self._prepare_logfile_and_summary()

    def _log(self, msg):
        with open(self.logfile, 'a') as f:
            f.write(msg)

    def _prepare_logfile_and_summary(self):
        # TrainerController.py:181-182: one event file per writer under <model_dir>/tensorboard/{train,validate}
        self.train_writer = tfevents.SummaryWriter(self.model_dir + '/tensorboard/train')
        self.val_writer = tfevents.SummaryWriter(self.model_dir + '/tensorboard/validate')
        self.logfile = self.model_dir + '/loss.csv'
        self._log('Network: %s\n' % self.network_name)
        self._log('Initial learning rate: %s\n' % self.learning_rate)
        self._log('Accuracy metric: %s\n' % self.accuracy_metric)
        self._log('Divergence weight: %s\n' % self.div_weight)
        stat_names = ','.join(self.loss_metrics.keys())
        self._log('epoch, %s, learning rate, elapsed (sec), best_model, benchmark_err, benchmark_rel_err, '
                  'benchmark_mse, benchmark_divloss\n' % stat_names)
        # source backup like TrainerController.py:196-206 (our package instead of ./Network)
        here = os.path.dirname(os.path.abspath(__file__))
        dest = os.path.join(self.model_dir, "backup_source")
        os.makedirs(dest, exist_ok=True)
        for fname in os.listdir(here):
            if fname.endswith(".py"):
                shutil.copy2(os.path.join(here, fname), os.path.join(dest, fname))

    def _update_summary_logging(self, epoch, results=None):
        """TrainerController.py:396-412: epoch-level scalars.  Train writer: '<name>/learning_rate' and every 'train_*' metric
        with the prefix stripped ('<name>/loss', '/accuracy', '/mse', '/div'); validate writer: the 'val_*' metrics likewise;
        step = the 0-based epoch.  ('l2_reg_loss' has neither prefix, so the reference does not log it -- nor do we.)
        results: metric name -> value (the rank-combined epoch means); defaults to this process's running means."""
        if results is None:
            results = dict((k, v.result()) for k, v in self.loss_metrics.items())
        self.train_writer.scalar("%s/learning_rate" % self.network_name, self.optimizer.lr, epoch)
        for k in self.loss_metrics:
            if k.startswith('train_'):
                self.train_writer.scalar("%s/%s" % (self.network_name, k.replace('train_', '')), results[k], epoch)
        for k in self.loss_metrics:
            if k.startswith('val_'):
                self.val_writer.scalar("%s/%s" % (self.network_name, k.replace('val_', '')), results[k], epoch)
        self.train_writer.flush()
        self.val_writer.flush()

    # ------------------------------------------------------------------ input staging
    def device_batches(self, dataset):
        """Iterate `dataset` ONE BATCH AHEAD of the consumer: while step k computes, batch k + 1 is copied to the device on a copy stream
        of its own -- non-blocking from the host loader's pinned ring (data.PatchHandler3D(..., pinned=True): a slot stays untouched
        until two more batches were requested, which covers the copy in flight), staged by torch otherwise.  Batches that already live on
        the device (DevicePatchHandler3D) pass through untouched.  What tf.data's prefetch-to-device does for the reference
        (PatchHandler3D.py:26-35 ends in .prefetch); FDN_H2D_PREFETCH=0 yields the dataset's own batches (one blocking copy per tensor
        inside train_step)."""
        if os.environ.get("FDN_H2D_PREFETCH", "1") in ("", "0"):
            yield from dataset
            return
        copy_stream = None

        def upload(batch):
            nonlocal copy_stream
            if all(isinstance(a, torch.Tensor) and a.is_cuda for a in batch):
                return batch, None
            if copy_stream is None:
                copy_stream = torch.cuda.Stream(device=self.device)
            # destinations come from the consumer stream's pool (and go back to it): the copy stream first waits for what that stream has
            # been given so far -- a recycled block may still be read by it -- which is at most the step before the one this batch feeds
            src = [a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)) for a in batch]
            dev = [torch.empty(t.shape, device=self.device, dtype=torch.float32) for t in src]
            copy_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(copy_stream):
                for d, t in zip(dev, src):
                    d.copy_(t, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(copy_stream)
            return tuple(dev), ev

        it = iter(dataset)
        try:
            nxt = upload(next(it))
        except StopIteration:
            return
        while nxt is not None:
            cur, ev = nxt
            if ev is not None:
                ev.synchronize()                           # `cur` has left its pinned slot (a copy-stream event: no wait for compute) before
            try:                                           # the loader is asked for more -- the ring's lifetime contract, kept by construction
                nxt = upload(next(it))                     # requested before the consumer gets `cur`: the copy runs beside its step
            except StopIteration:
                nxt = None
            if ev is not None:
                torch.cuda.current_stream().wait_event(ev)
            yield cur

    # ------------------------------------------------------------------ training loop
    def train_network(self, trainset, valset, n_epoch, testset=None, verbose=True):
        """TrainerController.py:263-343.  trainset/valset: iterables of 11-tuples (PatchHandler3D datasets)."""
        if self.model_dir is None:
            self.init_model_dir()
        is0 = parallel.rank() == 0
        if is0:
            print("==================== TRAINING =================")
            print('Learning rate %.7f' % self.optimizer.lr)
            print("Start training at %s - %s\n" % (time.ctime(), self.unique_model_name))
        start_time = time.time()
        previous_loss = np.inf
        total_batch_train = len(trainset) if hasattr(trainset, "__len__") else -1
        total_batch_val = len(valset) if hasattr(valset, "__len__") else -1
        for epoch in range(n_epoch):
            self.reset_metrics()
            start_loop = time.time()
            for i, data_pairs in enumerate(self.device_batches(trainset)):
                self.train_step(data_pairs)
                if verbose and is0:
                    print("\rEpoch %d Train batch %d/%d | loss: %.5f (%.1f %%) - %.1f secs" % (
                        epoch + 1, i + 1, total_batch_train, self.loss_metrics['train_loss'].result(),
                        self.loss_metrics['train_accuracy'].result(), time.time() - start_loop), end='')
            for i, data_pairs in enumerate(self.device_batches(valset)):
                self.test_step(data_pairs)
                if verbose and is0:
                    print("\rEpoch %d Validation batch %d/%d | loss: %.5f (%.1f %%) - %.1f secs" % (
                        epoch + 1, i + 1, total_batch_val, self.loss_metrics['val_loss'].result(),
                        self.loss_metrics['val_accuracy'].result(), time.time() - start_loop), end='')
            # data parallel: every rank saw a shard of each global batch -> combine (total, count) over ranks, so
            # loss.csv and the best-model decision do not depend on the world size
            res = dict((k, v.result_global()) for k, v in self.loss_metrics.items())
            message = "\rEpoch %d Train loss: %.5f (%.1f %%), Val loss: %.5f (%.1f %%) - %.1f secs" % (
                epoch + 1, res['train_loss'], res['train_accuracy'], res['val_loss'], res['val_accuracy'],
                time.time() - start_loop)
            loss_str = ','.join('%.5f' % res[k] for k in self.loss_metrics)
            log_line = "%d,%s,%.6f,%.1f" % (epoch + 1, loss_str, self.optimizer.lr, time.time() - start_loop)
            if is0:
                self._update_summary_logging(epoch, res)
            if res[self.accuracy_metric] < previous_loss:
                if is0:
                    self.save_best_model()
                previous_loss = res[self.accuracy_metric]
                message += ' **'
                log_line += ',**'
                if self.QUICKSAVE_ENABLED and testset is not None and is0:
                    ql, qa, qm, qd = self.quicksave(testset, epoch + 1)
                    message += ' Benchmark loss: %.5f (%.1f %%)' % (np.mean(ql), np.mean(qa))
                    log_line += ', %.7f, %.2f%%, %.7f, %.7f' % (np.mean(ql), np.mean(qa), np.mean(qm), np.mean(qd))
            if is0:
                print(message)
                self._log(log_line + "\n")
        if is0:
            el = time.time() - start_time
            hrs, mins = el // 3600, (el % 3600) // 60
            msg = "\nTraining %s completed! - name: %s" % (self.network_name, self.unique_model_name)
            msg += "\nTotal training time: %d hrs %d mins %d secs." % (hrs, mins, int(el - hrs * 3600 - mins * 60))
            msg += "\nFinished at %s\n==================== END TRAINING =================" % time.ctime()
            self._log(msg)
            print(msg)

    # ------------------------------------------------------------------ checkpoints
    def save_latest_model(self, epoch):
        """TrainerController.py:78-82 (defined but never called by the reference's loop; kept for API parity)."""
        if epoch > 0 and epoch % 10 == 0:
            self.model.save('%s-latest.h5' % self.model_path)
            print('Saving current model - %s\n' % time.ctime())

    def save_best_model(self):
        """TrainerController.py:347-363: '<dir>/<name>-best.h5' + optimizer.pkl = [iterations, m..., v...].

        ORDER of the m / v arrays: Keras writes `optimizer.get_weights()`, whose slots follow `model.trainable_variables`, i.e. the
        functional model's depth-sorted layer list -- NOT creation order: the phase convs precede the pc convs of equal depth and the
        three heads interleave (network.keras_layer_order).  The pickle is written (and read back by restore_model) in that order, so
        a restart file can travel between the reference and this implementation.  [TF]: the order is restated from Keras' published
        graph-sorting algorithm, TensorFlow is absent here; tests/test_tf_golden.py checks it against real variable names the day
        tests/golden/tf_golden.npz exists.  Several layers share a shape, so a wrong order cannot be detected from shapes alone."""
        self.model.save('%s-best.h5' % self.model_path)
        tv = self.model.trainable_variables
        sizes = [t.numel() for t in tv]
        shapes = [tuple(t.shape) for t in tv]
        m = [a.reshape(s) for a, s in zip(np.split(self.optimizer.m.cpu().numpy(), np.cumsum(sizes)[:-1]), shapes)]
        v = [a.reshape(s) for a, s in zip(np.split(self.optimizer.v.cpu().numpy(), np.cumsum(sizes)[:-1]), shapes)]
        order = self.model.keras_variable_order()
        weight_values = [np.int64(self.optimizer.iterations)] + [m[i] for i in order] + [v[i] for i in order]
        with open('%s/optimizer.pkl' % self.model_dir, 'wb') as f:
            pickle.dump(weight_values, f)
        # The pickle itself stays what Keras writes (a bare list).  A sidecar names the slots, so restore_model can map by NAME and
        # a file in another order (creation order: this repository before round 3) is re-ordered or refused instead of silently
        # mis-assigned -- conv3d / conv3d_2 etc. share shapes, so shapes cannot tell.
        names = self.model.trainable_variable_names()
        with open('%s/optimizer_order.txt' % self.model_dir, 'w') as f:
            f.write("# optimizer.pkl slot order: [iterations] + m slots + v slots, each in this variable order (Keras trainable_variables)\n")
            f.write("\n".join(names[i] for i in order) + "\n")

    def restore_model(self, old_model_dir, old_model_file):
        """TrainerController.py:365-394.  optimizer.pkl slots are in Keras trainable_variables order (see save_best_model)."""
        with open("%s/optimizer.pkl" % old_model_dir, 'rb') as f:
            opt_weights = pickle.load(f)
        tv = self.model.trainable_variables
        n = len(tv)
        if len(opt_weights) != 1 + 2 * n:
            raise ValueError("optimizer.pkl holds %d arrays, expected %d" % (len(opt_weights), 1 + 2 * n))
        order = self.model.keras_variable_order()
        names = self.model.trainable_variable_names()
        side = "%s/optimizer_order.txt" % old_model_dir
        legacy = os.environ.get("FDN_OPTIMIZER_PKL_ORDER", "")
        if os.path.exists(side):                              # written by save_best_model: map the slots by variable name
            listed = [l.strip() for l in open(side) if l.strip() and not l.startswith("#")]
            if sorted(listed) != sorted(names):
                raise ValueError("optimizer_order.txt names %d variables that are not this model's" % len(set(listed) ^ set(names)))
            order = [names.index(nm) for nm in listed]
        elif legacy == "creation":                            # a pickle of this repository before round 3 (no sidecar, creation order)
            order = list(range(n))
        elif legacy not in ("", "keras"):
            raise ValueError("FDN_OPTIMIZER_PKL_ORDER must be 'keras' or 'creation'")
        else:
            import warnings
            warnings.warn("optimizer.pkl without optimizer_order.txt: assuming Keras trainable_variables order (what the reference writes); "
                          "set FDN_OPTIMIZER_PKL_ORDER=creation for a file written by this repository before round 3")
        m_k, v_k = list(opt_weights[1:1 + n]), list(opt_weights[1 + n:])
        m, v = [None] * n, [None] * n
        for slot, i in enumerate(order):
            for name, src, dst in (("m", m_k, m), ("v", v_k, v)):
                if tuple(np.shape(src[slot])) != tuple(tv[i].shape):
                    raise ValueError("optimizer.pkl: %s slot %d has shape %s, expected %s (slots follow Keras trainable_variables order, "
                                     "see save_best_model)" % (name, slot, tuple(np.shape(src[slot])), tuple(tv[i].shape)))
                dst[i] = src[slot]
        self.optimizer.iterations = int(opt_weights[0])
        flat = lambda arrs: torch.from_numpy(np.concatenate([np.asarray(a, np.float32).reshape(-1) for a in arrs]))
        self.optimizer.m.copy_(flat(m))
        self.optimizer.v.copy_(flat(v))
        self.model.load_weights("%s/%s" % (old_model_dir, old_model_file))

    def quicksave(self, testset, epoch_nr):
        """TrainerController.py:415-454: predict the first benchmark batch, append to quicksave_<name>.h5."""
        from . import h5io
        for data_pairs in testset:
            inputs, hires, venc, mask = self._unpack(data_pairs)
            preds_t = self.model.forward(inputs)
            out, _ = ops.loss_metrics(preds_t, hires[0], hires[1], hires[2], mask, want_grad=False)
            break
        out = out.cpu().numpy()
        preds = preds_t.cpu().numpy()
        path = os.path.join(self.model_dir, "quicksave_%s.h5" % self.network_name)
        cols = []
        sv = lambda name, arr: cols.append((name, np.asarray(arr)))
        sv("epoch", np.asarray([epoch_nr]))
        pe = np.expand_dims(preds, 0)
        sv("u", pe[..., 0]); sv("v", pe[..., 1]); sv("w", pe[..., 2])
        if epoch_nr == 1:
            cpu = lambda t: t.cpu().numpy()
            sv("lr_u", cpu(inputs[0])); sv("lr_v", cpu(inputs[1])); sv("lr_w", cpu(inputs[2]))
            sv("hr_u", np.squeeze(cpu(hires[0]), -1)); sv("hr_v", np.squeeze(cpu(hires[1]), -1))
            sv("hr_w", np.squeeze(cpu(hires[2]), -1))
            sv("venc", cpu(venc)); sv("mask", cpu(mask))
        h5io.append_datasets(path, cols, compression='gzip')      # one pass over the file for all columns
        return out[:, 0], out[:, 1], out[:, 0], np.zeros_like(out[:, 0])
