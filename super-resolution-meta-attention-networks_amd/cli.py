"""train_sisr / eval_sisr entry points (same TOML schema and outputs as the reference's console scripts).

ref: Code/SISR/net_train.py:16-74 (experiment_setup), Code/SISR/training/training_handler.py:25-323
     (TrainingHandler: epoch loop, Y-PSNR validation, summary.csv, per-epoch checkpoints, early stopping),
     Code/SISR/net_eval.py:20-74 + Code/SISR/evaluation/standard_eval.py:217-319 (full_image_protocol).
Run:  python -m sisr_cli train --parameters cfg.toml      |      python -m sisr_cli eval --config cfg.toml
With WORLD_SIZE > 1 (torch.distributed.run) training is data parallel: every rank walks the same seeded
loader and takes its contiguous shard of each batch; rank 0 validates, logs and checkpoints.
"""
import argparse
import math
import os
import random
import time
from collections import defaultdict

import numpy as np
import torch

from . import metrics as M
from . import parallel
from .data import SuperResImages, sisr_data_setup
from .handlers import ModelInterface, create_dir_if_empty


def _load_toml(path):
    import tomli
    with open(path, 'rb') as f:
        return tomli.load(f)


def _dump_toml(d, path):
    def fmt(v):
        if isinstance(v, bool):
            return 'true' if v else 'false'
        if isinstance(v, (int, float)):
            return repr(v)
        if isinstance(v, (list, tuple)):
            return '[' + ', '.join(fmt(x) for x in v) + ']'
        return '"' + str(v).replace('\\', '\\\\').replace('"', '\\"') + '"'

    lines = []

    def emit(table, prefix):
        for k, v in table.items():
            if not isinstance(v, dict) and v is not None:
                lines.append(f'{k} = {fmt(v)}')
        for k, v in table.items():
            if isinstance(v, dict):
                lines.append(f'\n[{prefix + k}]')
                emit(v, prefix + k + '.')
    emit(d, '')
    with open(path, 'w') as f:
        f.write('\n'.join(lines) + '\n')


class TrainingHandler:
    def __init__(self, experiment_name, save_loc, model_params, data_params, gpu='off', sp_gpu=0, num_epochs=None,
                 continue_from_epoch=None, max_im_val=1.0, metrics=None, seed=8, epoch_cutoff=None,
                 early_stopping_patience=None, overwrite_data=False, **kwargs):
        self.rank, self.world, local = parallel.init_distributed()
        self.num_epochs, self.stop_patience, self.overwrite = num_epochs, early_stopping_patience, bool(overwrite_data)
        torch.manual_seed(seed)
        torch.cuda.manual_seed_all(seed)
        np.random.seed(seed)
        random.seed(seed)
        self.best_val_model_idx, self.best_val_model_psnr = 0, 0
        self.max_im_val = max_im_val
        self.metrics = list(metrics) if metrics is not None else None
        if self.world > 1:
            if os.environ.get("SISR_BENCH_SHARE_GPU"):  # rehearsal on a 1-GPU box: every rank on cuda:0
                local = 0
            gpu, sp_gpu = 'multi', local
        if self.rank != 0 and os.path.isdir(save_loc):
            pass  # every rank builds the interface; only rank 0 writes checkpoints / logs
        self.model = ModelInterface(save_loc, experiment_name, gpu=gpu, sp_gpu=int(sp_gpu), mode='train',
                                    new_params=model_params, load_epoch=continue_from_epoch)
        self.starting_epoch = self.model.model_epoch
        if self.starting_epoch > 0:
            self.starting_epoch += 1
        if epoch_cutoff is not None:
            self.num_epochs = epoch_cutoff - self.starting_epoch
        self.train_data, self.val_data = sisr_data_setup(scale=model_params['internal_params']['scale'],
                                                         device=self.model.model.device,
                                                         **self.model.configuration, **data_params)

    def train(self):
        losses = defaultdict(list)
        for batch in self.train_data:
            n = len(batch['tag'])
            if self.world > 1:
                batch = parallel.shard_batch(batch, self.rank, self.world)
            loss, _ = self.model.train_batch(**batch)
            if self.world > 1:  # logging only: the mean over the global batch from the ranks' shard means
                mine = len(batch['tag'])
                t = torch.tensor([float(loss) * mine if mine else 0.0, float(mine)], dtype=torch.float64)
                if torch.distributed.get_backend() == 'nccl':
                    t = t.to(self.model.model.device)
                torch.distributed.all_reduce(t)
                loss = np.float32(t[0].item() / n)
            losses['train-loss'].append(loss)
        losses['learning-rate'].append(self.model.get_learning_rate())
        self.model.epoch_end_calls()
        return losses

    def eval(self, epoch_idx):
        losses = defaultdict(list)
        for batch in self.val_data:
            y = batch['hr']
            rgb_out, ycbcr_out, loss, _ = self.model.net_run_and_process(**batch, request_loss=True)
            y_proc = self.model.colorspace_convert(y, colorspace='rgb')
            losses['val-loss'].append(loss)
            if self.metrics and 'PSNR' in self.metrics:
                for i in range(ycbcr_out.shape[0]):
                    losses['val-PSNR'].append(M.psnr(ycbcr_out[i, 0], y_proc[i, 0], max_value=self.max_im_val))
        return losses

    def run_experiment(self):
        import pandas as pd
        total = defaultdict(list)
        summary = os.path.join(self.model.logs, 'summary.csv')
        if self.starting_epoch != 0 and os.path.isfile(summary):
            old = pd.read_csv(summary)
            total = defaultdict(list, {k: list(old[k]) for k in old.columns})
        stale = 0
        for i, epoch_idx in enumerate(range(self.starting_epoch, self.starting_epoch + self.num_epochs)):
            t0 = time.time()
            print('Running epoch', epoch_idx)
            self.model.set_epoch(epoch_idx)
            if i == 0 and self.rank == 0:
                self.model.save(override=self.overwrite, dry_run=True)
            cur = dict(self.train())
            if self.rank != 0:
                iter(self.val_data)  # rank 0's validation iterator draws its base seed from the global torch RNG: keep
                #                      every rank's stream -- hence next epoch's shuffle -- identical
            if self.rank == 0:
                cur.update(self.eval(epoch_idx))
                val_psnr = np.mean(cur['val-PSNR']) if 'val-PSNR' in cur else float('nan')
                if val_psnr > self.best_val_model_psnr:
                    self.best_val_model_psnr, self.best_val_model_idx, stale = val_psnr, epoch_idx, 0
                else:
                    stale += 1
                for k, v in cur.items():
                    avg = np.nanmean(v)
                    total[k].append(0 if math.isnan(avg) else avg)
                total['epoch'].append(epoch_idx)
                self.model.save(override=self.overwrite)
                pd.DataFrame(total).to_csv(summary, index=False)
                print("Epoch {}:".format(epoch_idx), " ".join("{}_{:.4f}".format(k, np.mean(v)) for k, v in cur.items()),
                      "Epoch duration: {:.4f} seconds".format(time.time() - t0))
            if self.world > 1:
                flag = torch.tensor([1.0 if stale == self.stop_patience else 0.0], device=self.model.model.device)
                torch.distributed.broadcast(flag, src=0)
                if flag.item():
                    break
            elif self.stop_patience is not None and stale == self.stop_patience:
                print('Stopping model training, validation loss has plateaued.')
                break
        return total


def train_sisr(parameters, experiment_name=None, **overrides):
    """ref: net_train.py:29-74.  `parameters`: TOML path or an already-loaded dict."""
    params = _load_toml(parameters) if isinstance(parameters, str) else parameters
    params.setdefault('training', {}).update({k: v for k, v in overrides.items() if v is not None})
    if experiment_name is not None:
        params['experiment'] = experiment_name
    model = params['model']
    ml = model['internal_params'].get('metadata_list')
    if ml is not None:
        with open(ml) as f:
            model['internal_params']['metadata'] = [line.rstrip() for line in f]
    exp = TrainingHandler(experiment_name=params['experiment'], save_loc=params['experiment_save_loc'],
                          model_params=model, data_params=dict(params['data']), **params['training'])
    if exp.rank == 0:
        cont = params['training'].get('continue_from_epoch')
        _dump_toml(params, os.path.join(exp.model.base_folder,
                                        'config.toml' if cont is None else 'config_from_epoch_%s.toml' % cont))
        exp.model.save_metadata()
    return exp.run_experiment()


def eval_sisr(config=None, **kw):
    """ref: net_eval.py:64-74 / standard_eval.py full_image_protocol: per-image and average Y-PSNR CSVs."""
    import pandas as pd
    cfg = dict(_load_toml(config)) if config is not None else {}
    cfg.update({k: v for k, v in kw.items() if v is not None})
    out_dir = os.path.join(cfg.get('out_loc', '.'), cfg.get('results_name', 'delete_me'))
    os.makedirs(out_dir, exist_ok=True)
    scale = cfg.get('scale', 4)
    models = [ModelInterface(cfg['model_loc'], name, gpu='single' if cfg.get('gpu') else 'off',
                             sp_gpu=cfg.get('sp_gpu', 0), mode='eval', load_epoch=ep if ep in ('best', 'last') else int(ep),
                             scale=scale) for name, ep in cfg['model_and_epoch']]
    lr_dir = cfg['lr_dir']
    meta = cfg.get('metadata_file') or os.path.join(lr_dir, 'degradation_metadata.csv')
    if not os.path.isfile(meta):
        meta = None
    data = SuperResImages(lr_dir, cfg.get('hr_dir'), split='all' if cfg.get('full_directory') else (cfg.get('data_split') or 'eval'),
                          dataset=cfg.get('dataset_name'), scale=scale, degradation_metadata_file=meta,
                          recursive_search=bool(cfg.get('recursive')))
    loader = torch.utils.data.DataLoader(dataset=data, batch_size=cfg.get('batch_size', 1))
    rows = []
    for batch in loader:
        y_proc = ModelInterface.colorspace_convert(batch['hr'], colorspace='rgb')
        for m in models:
            rgb, ycbcr, _, secs = m.net_run_and_process(**batch, timing=cfg.get('time_models', True))
            for i, tag in enumerate(batch['tag']):
                rows.append({'Image_Name': tag, 'Model': m.experiment, 'PSNR': M.psnr(ycbcr[i, 0], y_proc[i, 0], 1),
                             'runtime': secs})
            if cfg.get('save_im'):
                from PIL import Image
                d = os.path.join(out_dir, m.experiment)
                os.makedirs(d, exist_ok=True)
                for i, tag in enumerate(batch['tag']):
                    Image.fromarray((rgb[i].transpose(1, 2, 0) * 255).round().astype(np.uint8)).save(
                        os.path.join(d, os.path.basename(tag)))
    df = pd.DataFrame(rows)
    mdir = os.path.join(out_dir, 'standard_metrics')
    create_dir_if_empty(mdir)
    df.to_csv(os.path.join(mdir, 'individual_metrics.csv'), index=False)
    avg = df.groupby('Model')[['PSNR', 'runtime']].mean().reset_index()
    avg.to_csv(os.path.join(mdir, 'average_metrics.csv'), index=False)
    return df, avg


def main(argv=None):
    ap = argparse.ArgumentParser(prog='sisr_cli')
    sub = ap.add_subparsers(dest='cmd', required=True)
    t = sub.add_parser('train')
    t.add_argument('--parameters', required=True)
    t.add_argument('--num_epochs', type=int)
    t.add_argument('--gpu', choices=['single', 'multi'])
    t.add_argument('--sp_gpu')
    t.add_argument('--experiment_name')
    t.add_argument('--seed', type=int, default=8)
    t.add_argument('--continue_from_epoch', type=int)
    t.add_argument('--overwrite_data', action='store_true', default=None)
    e = sub.add_parser('eval')
    e.add_argument('--config', required=True)
    a = ap.parse_args(argv)
    if a.cmd == 'train':
        kw = {k: v for k, v in vars(a).items() if k not in ('cmd', 'parameters', 'experiment_name')}
        train_sisr(a.parameters, experiment_name=a.experiment_name, **kw)
    else:
        eval_sisr(a.config)
