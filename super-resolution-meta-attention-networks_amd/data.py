"""Paired LR/HR image dataset producing the reference's batch-dict schema (host-side, CPU/PIL).

ref: Code/sr_tools/data_handler.py:147-528 (SuperResImages; __getitem__ :433-525, read_augmentation_list :62-144),
     Code/sr_tools/image_manipulation.py:233-257 (matched random crop, flip/rotate),
     Code/SISR/training/data_setup.py:9-125 (sisr_data_setup).
This is the caller side of the hot path: it is reproduced (same file ordering, same Python `random` call
order, same dict keys), not accelerated -- except `online_degradations`, whose blur + bicubic down-sampling of the HR
image runs on the device (degrade.py / csrc/degrade.hip; the random draws stay here, on numpy's global stream, in the
reference's order).  Options outside the in-scope configs (masks, CelebA attribute files, QPI group filters, Y-only
inputs) raise NotImplementedError.
"""
import glob
import json
import os
import random
import re

import numpy as np
import torch
from torch.utils.data import ConcatDataset, DataLoader, Dataset

DATA_SPLITS = {'celeba': {'train': (0, 162770), 'eval': (162770, 182637), 'test': (182637, 202599)},
               'div2k': {'train': (0, 800), 'eval': (800, 900)}, 'flickr2k': {'train': (0, 2650)}}


def image_names(folder, recursive=False):
    names = []
    for ext in ('*.jpg', '*.png', '*.bmp', '*.tif'):
        pat = os.path.join(folder, '**', ext) if recursive else os.path.join(folder, ext)
        names.extend(glob.glob(pat, recursive=recursive))
    names.sort()
    return names


def to_tensor(pil_image):
    """torchvision ToTensor for uint8 RGB: HWC uint8 -> CHW float32 / 255."""
    arr = np.asarray(pil_image)
    if arr.ndim == 2:
        arr = arr[:, :, None]
    return torch.from_numpy(np.ascontiguousarray(arr.transpose(2, 0, 1))).to(torch.float32).div(255)


def read_image(filename):
    import PIL.Image
    im = PIL.Image.open(filename)
    if im.mode in ('RGBA', 'L'):
        im = im.convert('RGB')
    return im


def center_crop(image, height, width):
    left = int(round((image.width - width) / 2.))
    top = int(round((image.height - height) / 2.))
    return image.crop((left, top, left + width, top + height))


def random_flip_rotate(lr, hr):
    """Three draws from `random` in the reference's order (hflip, vflip, rot90)."""
    hflip = random.random() < 0.5
    vflip = random.random() < 0.5
    rot90 = random.random() < 0.5

    def aug(img):
        if hflip:
            img = torch.flip(img, [2])
        if vflip:
            img = torch.flip(img, [1])
        if rot90:
            img = torch.transpose(img, 1, 2)
        return img
    return aug(lr), aug(hr)


def random_matched_crop(lr, hr, crop_size, scale):
    rnd_h = random.randint(0, max(0, lr.size()[1] - crop_size))
    rnd_w = random.randint(0, max(0, lr.size()[2] - crop_size))
    c_lr = lr[:, rnd_h:rnd_h + crop_size, rnd_w:rnd_w + crop_size]
    gh, gw = int(rnd_h * scale), int(rnd_w * scale)
    c_hr = hr[:, gh:gh + int(crop_size * scale), gw:gw + int(crop_size * scale)]
    return c_lr, c_hr


class DeviceTileSampler:
    """Training tiles cut on the GPU from a device-resident copy of the dataset.

    The reference decodes, augments and crops every sample on the host and ships float tensors through a
    DataLoader (data_handler.py:433-525).  A MI355X holds 288 GB: DIV2K + Flickr2K as planar fp32 is ~100 GB, so
    the decoded images are uploaded once and a step's batch is ONE gather kernel (`sisr_crop_augment`) -- no
    worker processes, no per-step host-to-device copy of pixels.  Results are exactly the reference's
    `random_flip_rotate` followed by `random_matched_crop`: the random draws are made here, on the host, with
    Python's `random` in the reference's call order (three `random()` for hflip / vflip / transpose, then two
    `randint` for the crop origin), only the pixel movement happens on the device.
    """

    def __init__(self, lr_images, hr_images, scale, crop, device, augment=True):
        """lr_images None: HR images only -- the LR side of every batch is synthesised on the device per access
        (DeviceTileLoader with `online_degradations`) and handed to sample_pairs."""
        from . import hip
        if not hr_images or (lr_images is not None and len(lr_images) != len(hr_images)):
            raise ValueError("need the same, non-zero number of LR and HR images")
        self.hip, self.scale, self.crop, self.augment = hip, int(scale), int(crop), bool(augment)
        self.device = torch.device(device)
        self.hr = [t.to(self.device, torch.float32).contiguous() for t in hr_images]
        self.lr = [t.to(self.device, torch.float32).contiguous() for t in lr_images] if lr_images is not None else []
        self.channels = self.hr[0].shape[0]
        for a, b in zip(self.lr, self.hr):
            if a.dim() != 3 or b.dim() != 3 or b.shape[1] != a.shape[1] * self.scale or b.shape[2] != a.shape[2] * self.scale:
                raise ValueError(f"HR {tuple(b.shape)} is not {self.scale}x LR {tuple(a.shape)}")
            if min(a.shape[1], a.shape[2]) < self.crop:
                raise ValueError(f"LR image {tuple(a.shape)} is smaller than the {self.crop}-pixel tile")

    def __len__(self):
        return len(self.hr)

    def draw(self, index, rng=random):
        """(top, left, hflip, vflip, transpose) for one sample, consuming `rng` exactly like the reference."""
        return self.draw_for(self.lr[index], rng)

    def draw_for(self, lr_image, rng=random):
        _, h, w = lr_image.shape
        hflip = vflip = rot = False
        if self.augment:
            hflip = rng.random() < 0.5
            vflip = rng.random() < 0.5
            rot = rng.random() < 0.5
        ah, aw = (w, h) if rot else (h, w)  # size of the augmented LR image
        top = rng.randint(0, max(0, ah - self.crop))
        left = rng.randint(0, max(0, aw - self.crop))
        return top, left, hflip, vflip, rot

    def sample(self, indices, rng=random):
        """-> (lr [B][C][crop][crop], hr [B][C][s*crop][s*crop]) fp32 tensors on the device."""
        return self.sample_pairs([(self.lr[i], self.hr[i]) for i in indices], rng)

    def sample_pairs(self, pairs, rng=random):
        """The same for explicit (LR, HR) device image pairs (contiguous planar fp32, HR = scale x LR): one gather per side."""
        hip, s = self.hip, self.scale
        B = len(pairs)
        rec_lr, rec_hr = [], []
        for lr_im, _ in pairs:
            top, left, hf, vf, rot = self.draw_for(lr_im, rng)
            _, h, w = lr_im.shape
            rec_lr.append([h, w, top, left, int(hf), int(vf), int(rot), 0])
            rec_hr.append([h * s, w * s, top * s, left * s, int(hf), int(vf), int(rot), 0])
        out = []
        for side, recs, c in ((0, rec_lr, self.crop), (1, rec_hr, self.crop * s)):
            ptrs = torch.tensor([p[side].data_ptr() for p in pairs], dtype=torch.int64).to(self.device)
            prm = torch.tensor(recs, dtype=torch.int32).to(self.device)
            dst = torch.empty((B, self.channels, c, c), device=self.device, dtype=torch.float32)
            hip.check(hip.lib().sisr_crop_augment(ptrs.data_ptr(), prm.data_ptr(), hip.ptr(dst), B, self.channels, c,
                                                  hip.stream()), "sisr_crop_augment")
            out.append(dst)
        return out[0], out[1]


class DeviceTileLoader:
    """Drop-in for the training DataLoader (`[data] device_tiles = true`): same batch dicts, same order, same
    random streams as `DataLoader(shuffle=True, num_workers=0)` over the same SuperResImages datasets -- but the
    images are decoded once, kept on the device, and every batch is cut there by DeviceTileSampler.

    Stream compatibility with torch's loader (so that a seeded run gives the same summary.csv either way): per
    epoch one int64 draw from the global torch RNG for the loader's base seed, one for the RandomSampler's seed,
    a `torch.randperm(n)` from a generator with that seed; per sample the five `random` draws of the sampler.
    """

    def __init__(self, datasets, batch_size, device, drop_last=False):
        self.batch_size, self.drop_last = int(batch_size), bool(drop_last)
        crops = {d.patch_crop for d in datasets}
        augs = {d.random_augment is not None for d in datasets}
        scales = {d.scale for d in datasets}
        if len(crops) != 1 or None in crops or len(augs) != 1 or len(scales) != 1:
            raise ValueError("device_tiles needs one common `crop` size, augmentation flag and scale on all training sets")
        lr, hr, self.tags, self.hr_tags, meta = [], [], [], [], []
        self.metadata_keys = datasets[0].metadata_keys
        online = {bool(getattr(d, 'online_degradations', False)) for d in datasets}
        if len(online) != 1:
            raise ValueError("device_tiles: either every training set synthesises its LR images (online_degradations) or none")
        self.online = online.pop()
        self.degraders = []  # per image: the degrader of its dataset (each dataset builds its own PCA basis, ref :228-238)
        for d in datasets:
            if d.hr_base is None:
                raise ValueError("device_tiles needs HR images (training sets)")
            if d.metadata_keys != self.metadata_keys:
                raise ValueError("training sets disagree on their metadata columns")
            if self.online:
                # ref data_handler.py:446-456: the LR image is synthesised from the HR one at every access.  Only the HR
                # images live on the device; each access blurs / (noises) / down-samples the WHOLE image there, as the
                # reference does before it crops, with the kernel (and noise) draws on the host in the reference's order
                for i in range(len(d)):
                    base_name = d.base_filenames[i]
                    hr.append(to_tensor(read_image(os.path.join(d.hr_base, base_name))))
                    self.tags.append(base_name)
                    self.hr_tags.append(base_name)
                    meta.append(d.metadata[i] if d.metadata is not None else None)  # file metadata, by HR name (ref :265)
                    self.degraders.append(d.degrader)
                continue
            for i in range(len(d)):
                base_name, image_name = d.base_filenames[i], d.lr_filenames[i]
                lr_im = read_image(os.path.join(d.lr_base, image_name))
                hr_im = read_image(os.path.join(d.hr_base, base_name))
                h, w = lr_im.height * d.scale, lr_im.width * d.scale
                if hr_im.width != w or hr_im.height != h:
                    hr_im = center_crop(hr_im, height=h, width=w)
                lr.append(to_tensor(lr_im))
                hr.append(to_tensor(hr_im))
                self.tags.append(image_name)
                self.hr_tags.append(base_name)
                meta.append(d.metadata[i] if d.metadata is not None else None)
        self.metadata = meta
        self.sampler = DeviceTileSampler(None if self.online else lr, hr, scale=scales.pop(), crop=crops.pop(), device=device,
                                         augment=augs.pop())
        self.dataset = self.sampler  # len(loader.dataset), as callers of a DataLoader expect

    def __len__(self):
        n = len(self.sampler)
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        n = len(self.sampler)
        torch.empty((), dtype=torch.int64).random_()                      # the loader iterator's base seed
        seed = int(torch.empty((), dtype=torch.int64).random_().item())   # RandomSampler's generator seed
        g = torch.Generator()
        g.manual_seed(seed)
        order = torch.randperm(n, generator=g).tolist()
        for start in range(0, n, self.batch_size):
            idx = order[start:start + self.batch_size]
            if len(idx) < self.batch_size and self.drop_last:
                return
            B = len(idx)
            zeros = torch.zeros(B, dtype=torch.int64)
            if self.online:
                pairs, codes, kernels = [], [], []
                for i in idx:  # per sample, in batch order: degrader draws (np.random), then flip / crop draws (random)
                    hr_full = self.sampler.hr[i]
                    lr_dev, code, kernel, (top, left, rh, rw) = self.degraders[i](hr_full)
                    hr_c = hr_full if (rh, rw) == tuple(hr_full.shape[1:]) else hr_full[:, top:top + rh, left:left + rw].contiguous()
                    pairs.append((lr_dev.contiguous(), hr_c))
                    codes.append(code.numpy() if self.metadata[i] is None else np.concatenate((self.metadata[i], code.numpy())))
                    kernels.append(kernel.numpy().squeeze())
                lr, hr = self.sampler.sample_pairs(pairs)
                yield {'lr': lr, 'hr': hr, 'tag': [self.tags[i] for i in idx], 'hr_tag': [self.hr_tags[i] for i in idx],
                       'mask': zeros.clone(), 'halfway_data': zeros.clone(), 'metadata': torch.from_numpy(np.stack(codes)),
                       'metadata_keys': _collate_keys(self.metadata_keys, B),
                       'blur_kernels': torch.from_numpy(np.stack(kernels))}
                continue
            lr, hr = self.sampler.sample(idx)
            if self.metadata[idx[0]] is not None:
                md = torch.from_numpy(np.stack([self.metadata[i] for i in idx]))
            else:
                md = zeros.clone()
            yield {'lr': lr, 'hr': hr, 'tag': [self.tags[i] for i in idx], 'hr_tag': [self.hr_tags[i] for i in idx],
                   'mask': zeros.clone(), 'halfway_data': zeros.clone(), 'metadata': md,
                   'metadata_keys': [tuple(k for _ in range(B)) for k in self.metadata_keys], 'blur_kernels': zeros.clone()}


def _collate_keys(keys, B):
    """What torch's default_collate makes of B copies of a dataset's metadata_keys: a string becomes a B-tuple of itself, a
    nested list of strings (the reference's kernel keys next to file metadata, see SuperResImages) a list of such tuples."""
    return [_collate_keys(k, B) if isinstance(k, (list, tuple)) else tuple(k for _ in range(B)) for k in keys]


def read_degradation_metadata(metadata_file, filenames):
    """CSV (index = image name) -> ({name: vector}, keys).  List columns (JSON) expand to repeated keys; integer
    columns are min-max normalised (QPI over the fixed 20..40 range)."""
    import pandas as pd
    keys = []
    data = pd.read_csv(metadata_file, header=0, index_col=0)
    for col in data:
        if data[col].dtype == object:
            data[col] = data[col].apply(json.loads)
            keys.extend([col.lower()] * len(data[col].iloc[0]))
        elif data[col].dtype == int:
            data[col] = data[col].astype(float)
            keys.append(col.lower())
            lo, hi = (20, 40) if col == 'QPI' else (data[col].min(), data[col].max())
            data[col] = (data[col] - lo) / (hi - lo)
        else:
            raise RuntimeError('Unidentified datatype in metadata file.')
    table = data.T.to_dict('list')
    out = {}
    for name in filenames:
        vec = []
        for v in table[name]:
            vec.extend(v) if isinstance(v, list) else vec.append(v)
        out[name] = np.array(vec)
    return out, keys


class SuperResImages(Dataset):
    def __init__(self, lr_dir=None, hr_dir=None, dataset=None, split=None, custom_split=None, recursive_search=False,
                 input='unmodified', colorspace='rgb', scale=4, degradation_metadata_file=None, metadata=None,
                 random_augments=None, random_crop=None, online_degradations=None, online_degradation_params=None,
                 **unsupported):
        super().__init__()
        if split not in ['train', 'eval', 'test', 'all', None]:
            raise RuntimeError('"Split" must be one of: train | eval | test | all | None')
        if input not in ['interp', 'unmodified']:
            raise RuntimeError('"lr_type" must be one of: interp | unmodified')  # ref: data_handler.py:193-194
        if 'rgb' not in colorspace:
            raise NotImplementedError('only RGB inputs are in scope (the Y-channel models SRCNN / VDSR are not)')
        if input == 'interp' and online_degradations:
            raise NotImplementedError("online degradations produce low-resolution images: input = 'interp' (LR images stored "
                                      "at HR size, SPARNet) does not go with them")
        self.lr_type = input
        for k in ('mask_data', 'halfway_data', 'blacklist', 'data_attributes', 'image_shortlist',
                  'legacy_blur_kernels', 'request_crops', 'group_select', 'attribute_amplification'):
            if unsupported.get(k):
                raise NotImplementedError(f'data option {k!r} is outside the HIP hot-path scope')
        self.scale, self.patch_crop, self.random_augment = scale, random_crop, random_augments
        self.lr_base, self.hr_base = lr_dir, hr_dir
        self.online_degradations = bool(online_degradations)
        if self.online_degradations:
            # ref: data_handler.py:222-238 -- LR images are synthesised from the HR ones; the degrader's PCA basis is
            # built here, from 30 000 random kernels of numpy's global stream, before any file is listed
            if hr_dir is None:
                raise RuntimeError('Cannot synthesize LR images without specifying HR images.')
            from . import degrade
            self.degrader = degrade.OnlineDegrader(scale=scale, **(online_degradation_params or {}))
            self.lr_base, lr_dir = None, hr_dir
        groups = {}
        for f in image_names(lr_dir, recursive_search):
            rel = os.path.relpath(f, lr_dir)
            parts = re.split(r"_q(.*)(?=\.)", rel)
            base = parts[0] + parts[2] if len(parts) > 1 else parts[0]
            groups.setdefault(base, []).append(rel)
        items = list(groups.items())
        if custom_split is not None or (split != 'all' and len(items) != 1):
            start, end = custom_split if custom_split is not None else DATA_SPLITS[dataset][split]
            items = items[start:end]
        self.lr_filenames = [f for _, fs in items for f in fs]
        self.base_filenames = [b for b, fs in items for _ in fs]
        if not self.lr_filenames:
            raise RuntimeError('No images were supplied or all images were filtered out!')
        self.metadata, self.metadata_keys = None, []
        if degradation_metadata_file is not None:
            # ref :264-287: rows are looked up by LR file name, or -- with online degradations, where there are no LR files --
            # by the HR (base) name
            names = self.base_filenames if self.online_degradations else self.lr_filenames
            table, self.metadata_keys = read_degradation_metadata(degradation_metadata_file, names)
            self.metadata = [table[n] for n in names]
        if self.online_degradations:
            kernel_keys = ['blur_kernel'] * self.degrader.para_in
            if len(self.metadata_keys) == 0:
                self.metadata_keys = kernel_keys  # ref :293-295
            else:
                # ref :296-297 APPENDS the list as ONE element (keys = [k1, .., kn, ['blur_kernel', ...]]): a sample's vector
                # is file metadata + kernel code, but the kernel code's entries have no key of their own, so a model that
                # selects 'blur_kernel' by key finds none of them.  Kept as the reference has it (traps).
                self.metadata_keys = list(self.metadata_keys) + [kernel_keys]
        self.image_count = len(self.lr_filenames)
        print('Initialized %s data with %d image%s.' % (dataset if dataset is not None else 'image', self.image_count,
                                                        's' if self.image_count > 1 else ''))

    def __len__(self):
        return self.image_count

    def __getitem__(self, index):
        base_name, image_name = self.base_filenames[index], self.lr_filenames[index]
        if self.online_degradations:
            return self._degraded_item(base_name, index)
        lr_im = read_image(os.path.join(self.lr_base, image_name))
        metadata = self.metadata[index] if self.metadata is not None else np.array(0)
        if self.hr_base is not None:
            hr_im = read_image(os.path.join(self.hr_base, base_name))
            if self.lr_type == 'interp':  # ref: data_handler.py:473-476: the LR images are stored already interpolated
                h, w = lr_im.height, lr_im.width
            else:
                h, w = lr_im.height * self.scale, lr_im.width * self.scale
            if hr_im.width != w or hr_im.height != h:
                hr_im = center_crop(hr_im, height=h, width=w)
            hr_im = to_tensor(hr_im)
        else:
            hr_im = np.array(0)
        lr_im = to_tensor(lr_im)
        if self.random_augment is not None:
            lr_im, hr_im = random_flip_rotate(lr_im, hr_im)
        if self.patch_crop is not None:
            lr_im, hr_im = random_matched_crop(lr_im, hr_im, crop_size=self.patch_crop, scale=self.scale)
        return {'lr': lr_im, 'hr': hr_im, 'tag': image_name, 'hr_tag': base_name, 'mask': np.array(0),
                'halfway_data': np.array(0), 'metadata': metadata, 'metadata_keys': self.metadata_keys,
                'blur_kernels': np.array(0)}

    def _degraded_item(self, base_name, index=None):
        """ref: data_handler.py:446-456 + the common tail of __getitem__: blur kernel drawn on the host (np.random), blur +
        quantisation + PIL bicubic on the device, kernel code as the sample's metadata, full kernel as 'blur_kernels'."""
        if not torch.cuda.is_available():
            raise RuntimeError('online degradations run on a HIP device (no CPU path in this package)')
        hr_im = to_tensor(read_image(os.path.join(self.hr_base, base_name)))
        lr_dev, code, kernel, (top, left, rh, rw) = self.degrader(hr_im.cuda())
        lr_im = lr_dev.cpu()
        hr_im = hr_im[:, top:top + rh, left:left + rw]  # center_crop to LR size x scale (ref :470-476)
        if self.random_augment is not None:
            lr_im, hr_im = random_flip_rotate(lr_im, hr_im)
        if self.patch_crop is not None:
            lr_im, hr_im = random_matched_crop(lr_im, hr_im, crop_size=self.patch_crop, scale=self.scale)
        metadata = code.numpy()
        if self.metadata is not None and index is not None:  # ref :451-452: file metadata first, then the kernel code
            metadata = np.concatenate((self.metadata[index], metadata))
        return {'lr': lr_im, 'hr': hr_im, 'tag': base_name, 'hr_tag': base_name, 'mask': np.array(0),
                'halfway_data': np.array(0), 'metadata': metadata, 'metadata_keys': self.metadata_keys,
                'blur_kernels': kernel.numpy().squeeze()}


def sisr_data_setup(training_sets, eval_sets, batch_size=16, eval_batch_size=1, dataloader_threads=8,
                    drop_last_training_batch=False, device_tiles=False, device=None, **common):
    """TOML [data] block -> (train DataLoader, val DataLoader).  ref: training/data_setup.py:9-125
    `device_tiles = true` (not a reference option): training batches are cut on the GPU from a device-resident copy
    of the training images (DeviceTileLoader) instead of by DataLoader workers."""
    common = {k: v for k, v in common.items() if k in ('scale', 'input', 'colorspace')}

    def setup(ds, split):
        custom = None
        if ds.get('cutoff') is not None:
            custom = ds['cutoff'] if isinstance(ds['cutoff'], list) else (0, ds['cutoff'])
        elif ds.get('name') is None:
            split = 'all'
        meta_file = ds.get('degradation_metadata') or ds.get('qpi_values')
        if meta_file == 'on_site':
            meta_file = os.path.join(ds['lr'], 'degradation_metadata.csv')
            if not os.path.isfile(meta_file):
                meta_file = os.path.join(ds['lr'], 'qpi_slices.csv')
        extra = {k: ds.get(k) for k in ('online_degradations', 'online_degradation_params', 'image_shortlist',
                                        'legacy_blur_kernels', 'request_crops', 'attribute_amplification')}
        return SuperResImages(lr_dir=ds['lr'], hr_dir=ds.get('hr'), dataset=ds.get('name'), split=split,
                              custom_split=custom, degradation_metadata_file=meta_file, metadata=ds.get('metadata'),
                              random_crop=ds.get('crop'), random_augments=ds.get('random_augment'),
                              recursive_search=bool(ds.get('recursive_search')), **extra, **common)

    train = [setup(d, 'train') for d in training_sets.values()]
    val = [setup(d, 'eval') for d in eval_sets.values()]
    val = val[0] if len(val) == 1 else ConcatDataset(val)
    if device_tiles:
        if device is None or not torch.cuda.is_available():
            raise RuntimeError("device_tiles needs a HIP device")
        return (DeviceTileLoader(train, batch_size, device, drop_last=drop_last_training_batch),
                DataLoader(dataset=val, batch_size=eval_batch_size))
    if any(getattr(d, 'online_degradations', False) for d in train):
        dataloader_threads = 0  # the degrader launches HIP kernels: it cannot run in forked DataLoader workers
    train = train[0] if len(train) == 1 else ConcatDataset(train)
    train_loader = DataLoader(dataset=train, batch_size=batch_size, shuffle=True, num_workers=dataloader_threads,
                              pin_memory=torch.cuda.is_available(), drop_last=drop_last_training_batch)
    return train_loader, DataLoader(dataset=val, batch_size=eval_batch_size)
