"""Batch form of ctpn/demo.py: a directory of images in, `res_<stem>.txt` (+ annotated images) out,
with the same per-image arithmetic as demo.ctpn() (reference ctpn/demo.py:55-68) but

  * image sizes come from the file headers, so the batches (grouped by the size after resize_im) are known before a pixel
    is decoded; decode (Pillow, releases the GIL) + resize_im on the GPU (ctpn_resize) of batch k+1 run on a host thread
    pool while batch k is on the GPU (the reference decodes with cv2.imread on the one Python thread, demo.py:59),
  * with --decode gpu the JPEG files are decoded and resized on the device (ctpn_decode_jpeg_batch: entropy decoding on the library's host
    pool, everything after it as HIP kernels), batches grouped by file size; other formats keep the host decoder,
  * batches go through ctpn_detect_submit / ctpn_detect_collect (the reference asserts batch == 1,
    lib/rpn_msr/proposal_layer_tf.py:51), software-pipelined over the ctx's two slots; the ctx is sized ONCE for the largest
    batch / shape of the run (growing it mid-run would destroy the slot that still holds an uncollected batch).

    python -m ctpn_amd.ctpn.demo_batch --input data/demo --out data/results --batch 32 [--mode O] [--synthetic 0] [--no-images]
"""
from __future__ import print_function

import argparse
import glob
import os
import sys
import time

import numpy as np

_PKG_PARENT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _PKG_PARENT not in sys.path:
    sys.path.insert(0, _PKG_PARENT)

import ctpn_amd  # noqa: E402,F401
from ctpn_amd.ctpn import demo as D  # noqa: E402
from ctpn_amd import _binding as B  # noqa: E402
from ctpn_amd.lib.networks.factory import get_network  # noqa: E402
from ctpn_amd.lib.fast_rcnn.config import cfg, cfg_from_file  # noqa: E402
from ctpn_amd.lib.fast_rcnn.test import _scale_for  # noqa: E402
from ctpn_amd.lib.utils import image as imutil  # noqa: E402
from ctpn_amd.lib.text_connector.text_connect_cfg import Config as TextLineCfg  # noqa: E402


def list_images(path):
    if os.path.isdir(path):
        names = []
        for ext in ("*.png", "*.jpg", "*.jpeg", "*.bmp"):
            names += glob.glob(os.path.join(path, ext))
        return sorted(names)
    return sorted(glob.glob(path))


def image_size(path):
    """(h, w) of imread(path) from the file header only (no pixel decode; a JPEG's EXIF orientation taken into account like cv2.imread)."""
    return imutil.image_size(path)


def plan(names, batch):
    """-> (jobs, singles, shapes): jobs = [(resized (h, w), [names])] batched by the shape after resize_im; singles = images whose
    second rescale (TEST.SCALES / MAX_SIZE, test.py:17-24) is not the identity (they take the single-image blob path)."""
    from ctpn_amd._binding import resize_dims
    groups, singles, shapes = {}, [], {}
    for name in names:
        h, w = image_size(name)
        f = D.resize_factor((h, w), TextLineCfg.SCALE, TextLineCfg.MAX_SCALE)
        rs = (h, w) if f == 1.0 else resize_dims(h, w, f, f)
        shapes[name] = rs
        s2 = _scale_for(rs)
        if int(round(rs[0] * s2)) == rs[0] and int(round(rs[1] * s2)) == rs[1]:
            groups.setdefault(rs, []).append(name)
        else:
            singles.append(name)
    jobs = []
    for shape, members in sorted(groups.items()):
        for i in range(0, len(members), batch):
            jobs.append((shape, members[i:i + batch]))
    return jobs, singles, shapes


def _check_uint8_feed_config():
    """The batched path feeds uint8 images; the library subtracts the reference's PIXEL_MEANS (lib/fast_rcnn/config.py:200) inside its first
    kernel (csrc/layers.hip), compiled in. The reference subtracts cfg.PIXEL_MEANS at run time (lib/fast_rcnn/test.py:7-11), so an edited
    value must not be ignored silently: it is an error here (the single-image path, lib/fast_rcnn/test.py, subtracts cfg.PIXEL_MEANS in
    Python and takes any value)."""
    built = np.array([102.9801, 115.9465, 122.7717])
    if not np.allclose(np.asarray(cfg.PIXEL_MEANS, np.float64).reshape(-1), built, rtol=0, atol=1e-6):
        raise ValueError("cfg.PIXEL_MEANS = %s, but the uint8 batch feed of libctpn_hip.so subtracts %s in its first kernel; use ctpn/demo.py's "
                         "float-blob path for other means" % (np.asarray(cfg.PIXEL_MEANS).reshape(-1).tolist(), built.tolist()))
    # ctpn_detect / ctpn_detect_submit run the proposal layer with the reference's TEST values (csrc/ctpn_api.hip: 12000, 1000, 0.7, 8) --
    # ctpn_proposals, the single-image seam, takes them as arguments (lib/fast_rcnn/test.py)
    t = cfg.TEST
    got = (int(t.RPN_PRE_NMS_TOP_N), int(t.RPN_POST_NMS_TOP_N), float(t.RPN_NMS_THRESH), float(t.RPN_MIN_SIZE))
    if got != (12000, 1000, 0.7, 8.0):
        raise ValueError("cfg.TEST.RPN_PRE_NMS_TOP_N / RPN_POST_NMS_TOP_N / RPN_NMS_THRESH / RPN_MIN_SIZE = %r, but the batched path "
                         "(ctpn_detect) runs the proposal layer with (12000, 1000, 0.7, 8); ctpn/demo.py's single-image path takes any values" % (got,))


def _load(name):
    img = imutil.imread(name)
    return D.resize_im(img, scale=TextLineCfg.SCALE, max_scale=TextLineCfg.MAX_SCALE)


def _decode_into(name, shm_name, batch_shape, index):
    """Worker PROCESS: decode `name` (Pillow) straight into slot `index` of the shared uint8 batch buffer if the file already has the
    batch's shape (resize_im's factor is 1: the benchmark's case); otherwise hand the decoded image back for the parent's GPU resize.
    Threads do not scale here -- Pillow's RGB conversion and the BGR copy hold the GIL (measured, profiles/r04_decode_throughput_*.json:
    1830 JPEG/s on 32 threads against 330 on one) -- processes do."""
    rgb = imutil.open_rgb(name)
    if rgb.shape[:2] != tuple(batch_shape[1:3]):
        return np.ascontiguousarray(rgb[:, :, ::-1])
    np.ndarray(batch_shape, np.uint8, buffer=_attach(shm_name).buf)[index] = rgb[:, :, ::-1]
    return None


_SHM = {}


def _attach(shm_name):
    """A worker maps each shared batch buffer ONCE (attaching per task costs an mmap of the whole batch, its page faults and a round trip to
    multiprocessing's resource tracker: measured 1343 -> see profiles/r04_decode_throughput.json)."""
    shm = _SHM.get(shm_name)
    if shm is None:
        from multiprocessing import shared_memory
        if len(_SHM) > 8:
            for old in list(_SHM.values()):
                old.close()
            _SHM.clear()
        shm = _SHM[shm_name] = shared_memory.SharedMemory(name=shm_name)
    return shm


def decode_pool(procs):
    """A pool of `procs` decode worker processes (spawned: the parent holds a HIP context), warmed up so that a timed run does not pay for
    their start-up (~1.5 s for 32 workers). Pass it to run(decode_pool=...); the caller shuts it down."""
    import multiprocessing as mp
    from concurrent.futures import ProcessPoolExecutor
    pool = ProcessPoolExecutor(max_workers=procs, mp_context=mp.get_context("spawn"))
    list(pool.map(_warm, range(4 * procs)))
    return pool


def _warm(i):
    import PIL.Image  # noqa: F401
    time.sleep(0.02)       # long enough that every worker of the pool takes some
    return i


def _is_jpeg(name):
    return name.lower().endswith((".jpg", ".jpeg"))


def _read(name):
    with open(name, "rb") as f:
        return f.read()


def _run_gpu(net, names, out_dir, batch, mode, write_images, log, read_threads=8):
    """decode='gpu': the JPEG files of the run are decoded AND resized on the device (ctpn_decode_jpeg_batch: Huffman decoding on the ctx's
    C++ worker pool, IDCT / chroma upsampling / colour conversion / cv2.resize as HIP kernels in the ctx's copy queue, ordered against the
    forward by events) -- neither the file bytes nor the pixels pass through Python, and the pixels never exist on the host unless
    annotated images are asked for. Batches are grouped by FILE size (as cv2.imread returns it: EXIF orientation applied), chroma layout
    and orientation, all read from the headers (one size, one resize factor, one network shape per batch). PNG files are decoded by the library too, on the host by the nature of the format
    (ctpn_decode_png_files: inflate + row filters, one file per C++ thread, straight into the batch buffer that ctpn_detect_submit copies to
    the device). Files neither decoder takes (CMYK / arithmetic-coded / truncated JPEG, 16-bit PNG, other formats) go through Pillow
    (lib/utils/image.py), batched the same way; the result files are the same whichever decoder a file went through."""
    from ctpn_amd._binding import resize_dims
    mode = mode or cfg.TEST.DETECT_MODE
    os.makedirs(out_dir, exist_ok=True)
    groups, singles = {}, []
    t_plan = time.time()
    probed = B.jpeg_probe_files(names, read_threads)                          # the header scan: one call, C++ threads
    pngs = [i for i, nm in enumerate(names) if probed[i, 0] == 0 and nm.lower().endswith(".png")]
    png_info = dict(zip(pngs, B.png_probe_files([names[i] for i in pngs], read_threads).tolist())) if pngs else {}
    PNG, OTHER = (-1, 0), (0, 0)                                              # layouts of the files the JPEG decoder does not take
    for i, (name, pr) in enumerate(zip(names, probed.tolist())):
        if pr[0] > 0:
            (h, w), layout = (pr[0], pr[1]), (pr[2], pr[3])
        elif png_info.get(i, [0])[0] > 0:
            (h, w), layout = tuple(png_info[i][:2]), PNG
        else:
            (h, w), layout = image_size(name), OTHER
        f = D.resize_factor((h, w), TextLineCfg.SCALE, TextLineCfg.MAX_SCALE)
        rs = (h, w) if f == 1.0 else resize_dims(h, w, f, f)
        s2 = _scale_for(rs)
        if int(round(rs[0] * s2)) == rs[0] and int(round(rs[1] * s2)) == rs[1]:
            groups.setdefault((h, w, layout), (f, rs, []))[2].append(name)
        else:
            singles.append(name)
    jobs = []
    for (h, w, layout), (f, rs, members) in sorted(groups.items()):
        for i in range(0, len(members), batch):
            jobs.append(((h, w), "jpg" if layout[0] > 0 else ("png" if layout == PNG else "host"), f, rs, members[i:i + batch]))
    if jobs:
        net.ensure_capacity(max(len(j[4]) for j in jobs), max(j[3][0] for j in jobs), max(j[3][1] for j in jobs))
    results, meta, stats = {}, {}, {"gpu": 0, "png": 0, "host": 0}
    # PNG batches are decoded ONE JOB AHEAD on a helper thread (the C++ decode threads hang off that call; ctypes releases the GIL), so that
    # batch k + 1 inflates while batch k is submitted and batch k - 1 collected. Three batch buffers per shape in a ring: when batch k + 1
    # starts decoding, batch k sits decoded in its buffer and batch k - 1 may still be on its way to the device. THREE slots whatever the
    # run's shapes: a slot whose shape changes is replaced (a directory of a thousand PNG sizes does not keep a thousand batch buffers).
    from concurrent.futures import ThreadPoolExecutor
    png_bufs, png_ahead, png_pool = {}, {}, ThreadPoolExecutor(max_workers=1)

    def png_decode(k):
        (h, w), _, f, rs, members = jobs[k]
        shape = (len(members), h, w, 3)
        if k % 3 not in png_bufs or png_bufs[k % 3].shape != shape:
            png_bufs[k % 3] = np.empty(shape, np.uint8)
        imgs = B.decode_png_files(members, h, w, read_threads, out=png_bufs[k % 3])      # files read, inflated and unfiltered on C++ threads
        if f != 1.0:
            imgs = B.resize_linear(imgs, f, f)
        assert tuple(imgs.shape[1:3]) == tuple(rs), (imgs.shape, rs)
        return imgs

    def png_prefetch(k):
        if k < len(jobs) and jobs[k][1] == "png" and k not in png_ahead:
            png_ahead[k] = png_pool.submit(png_decode, k)
    t0 = time.time()
    t_plan = t0 - t_plan
    pending = None

    def emit(nm):
        img, scale = meta.pop(nm)
        if write_images:
            D.draw_boxes(img.copy(), nm, results[nm], scale, out_dir)
        else:
            base = os.path.basename(nm)
            B.write_result_file(os.path.join(out_dir, 'res_{}.txt'.format(base.split('.')[0])), results[nm], scale)

    def collect(job):                                  # ... and its result files are written here, while the next batch is on the GPU
        slot, members = job
        for nm, recs in zip(members, net.ctx.detect_collect(slot, mode=mode, line_capacity=1024)):
            results[nm] = recs
        for nm in members:
            emit(nm)

    png_prefetch(0)
    for k, ((h, w), kind, f, rs, members) in enumerate(jobs):
        png_prefetch(k + 1)
        imgs = None
        if kind == "jpg":
            try:
                ptr, shape = net.ctx.decode_jpeg_files(members, h, w, f, f)      # files read + entropy-decoded on the library's pool
                assert tuple(shape[1:]) == tuple(rs), (shape, rs)
                net.ctx.detect_submit(device_ptr=ptr, shape=shape, slot=k & 1)
                stats["gpu"] += len(members)
                if write_images:
                    imgs = net.ctx.jpeg_batch_fetch(ptr, shape)
            except B.CtpnError as e:                                       # e.g. damaged entropy data: the host decoder's call
                if e.code not in (B.CTPN_ERR_UNSUPPORTED, -1):
                    raise
                kind = "host"
        elif kind == "png":
            try:
                imgs = png_ahead.pop(k).result()
                net.ctx.detect_submit(images=imgs, slot=k & 1)
                stats["png"] += len(members)
            except B.CtpnError as e:
                if e.code not in (B.CTPN_ERR_UNSUPPORTED, -1):
                    raise
                kind = "host"
        if kind == "host":
            imgs = np.stack([_load(nm)[0] for nm in members])
            net.ctx.detect_submit(images=imgs, slot=k & 1)
            stats["host"] += len(members)
        for i, nm in enumerate(members):
            meta[nm] = (imgs[i] if imgs is not None and write_images else None, f)
        if pending is not None:
            collect(pending)
        pending = (k & 1, members)
    if pending is not None:
        collect(pending)
    png_pool.shutdown()
    for nm in singles:
        img, scale = _load(nm)
        from ctpn_amd.lib.fast_rcnn.test import test_ctpn
        from ctpn_amd.lib.text_connector.detectors import TextDetector
        scores, boxes = test_ctpn(None, net, img)
        results[nm] = TextDetector().detect(boxes, scores[:, np.newaxis], img.shape[:2])
        meta[nm] = (img, scale)
        emit(nm)
    dt = time.time() - t0
    log('Detection of {:d} images in {:d} batches took {:.3f}s, result files included, after a header scan of {:.3f}s ({:.1f} images/s; {:d} decoded on the device, {:d} PNG files by the library, {:d} on the host)'.format(
        len(names), len(jobs) + len(singles), dt, t_plan, len(names) / max(dt, 1e-9), stats["gpu"], stats["png"], stats["host"] + len(singles)))
    return results


def run(net, names, out_dir, batch=32, mode=None, write_images=True, log=print, decode_threads=8, decode_procs=0, decode_pool=None, decode="host"):
    """-> {image name: (M,9) records}. decode_procs > 0 (or a warm decode_pool): decode in worker processes writing into shared-memory batch
    buffers (one batch ahead of the GPU) instead of on the thread pool. decode='gpu': JPEG decode + resize_im on the device (_run_gpu)."""
    from concurrent.futures import ThreadPoolExecutor
    _check_uint8_feed_config()
    if decode == "gpu":
        return _run_gpu(net, names, out_dir, batch, mode, write_images, log, read_threads=decode_threads)
    if decode_procs > 0 or decode_pool is not None:
        return _run_procs(net, names, out_dir, batch, mode, write_images, log, decode_procs, decode_pool)
    mode = mode or cfg.TEST.DETECT_MODE
    os.makedirs(out_dir, exist_ok=True)
    jobs, singles, _ = plan(names, batch)
    if jobs:      # one ctx for the whole run: largest batch x largest shape
        net.ensure_capacity(max(len(m) for _, m in jobs), max(s[0] for s, _ in jobs), max(s[1] for s, _ in jobs))
    results, meta = {}, {}
    t0 = time.time()
    pending = None

    def collect(job):
        slot, members = job
        lines = net.ctx.detect_collect(slot, mode=mode, line_capacity=1024)
        for nm, recs in zip(members, lines):
            results[nm] = recs

    with ThreadPoolExecutor(max_workers=max(1, decode_threads)) as pool:
        def decode(members):
            return [pool.submit(_load, nm) for nm in members]
        ahead = decode(jobs[0][1]) if jobs else []
        for k, (shape, members) in enumerate(jobs):
            loaded = [f.result() for f in ahead]
            ahead = decode(jobs[k + 1][1]) if k + 1 < len(jobs) else []      # next batch decodes while this one is on the GPU
            for nm, (img, scale) in zip(members, loaded):
                assert img.shape[:2] == tuple(shape), (nm, img.shape, shape)
                meta[nm] = (img, scale)
            net.ctx.detect_submit(images=np.stack([img for img, _ in loaded]), slot=k & 1)
            if pending is not None:
                collect(pending)
            pending = (k & 1, members)
        if pending is not None:
            collect(pending)
        for nm, fut in zip(singles, [pool.submit(_load, nm) for nm in singles]):
            meta[nm] = fut.result()
    for nm in singles:
        img, scale = meta[nm]
        from ctpn_amd.lib.fast_rcnn.test import test_ctpn
        from ctpn_amd.lib.text_connector.detectors import TextDetector
        scores, boxes = test_ctpn(None, net, img)
        results[nm] = TextDetector().detect(boxes, scores[:, np.newaxis], img.shape[:2])
    dt = time.time() - t0
    for nm in names:
        img, scale = meta[nm]
        if write_images:
            D.draw_boxes(img.copy(), nm, results[nm], scale, out_dir)
        else:
            base = os.path.basename(nm)
            B.write_result_file(os.path.join(out_dir, 'res_{}.txt'.format(base.split('.')[0])), results[nm], scale)
    log('Detection of {:d} images in {:d} batches took {:.3f}s ({:.1f} images/s)'.format(len(names), len(jobs) + len(singles), dt, len(names) / max(dt, 1e-9)))
    return results


def _run_procs(net, names, out_dir, batch, mode, write_images, log, procs, pool=None):
    from multiprocessing import shared_memory
    own_pool = pool is None
    if own_pool:
        pool = decode_pool(procs)
    mode = mode or cfg.TEST.DETECT_MODE
    os.makedirs(out_dir, exist_ok=True)
    jobs, singles, _ = plan(names, batch)
    results, meta = {}, {}
    if jobs:
        net.ensure_capacity(max(len(m) for _, m in jobs), max(s[0] for s, _ in jobs), max(s[1] for s, _ in jobs))
    nbytes = max([len(m) * s[0] * s[1] * 3 for s, m in jobs] + [1])
    NB = 4                                                                                  # batches k + 1, k + 2 decode, k is on the GPU, k - 1's pixels are still referenced
    shms = [shared_memory.SharedMemory(create=True, size=nbytes) for _ in range(NB)]
    t0 = time.time()
    try:
        if True:
            def decode(k):
                shape, members = jobs[k]
                bshape = (len(members), shape[0], shape[1], 3)
                return bshape, [pool.submit(_decode_into, nm, shms[k % NB].name, bshape, i) for i, nm in enumerate(members)]
            ahead = {k: decode(k) for k in range(min(2, len(jobs)))}
            pending = None
            for k, (shape, members) in enumerate(jobs):
                bshape, futs = ahead.pop(k)
                arr = np.ndarray(bshape, np.uint8, buffer=shms[k % NB].buf)
                for i, (nm, f) in enumerate(zip(members, futs)):
                    back = f.result()
                    if back is not None:                                   # not at the batch shape yet: resize_im on the GPU, in the parent
                        img, scale = D.resize_im(back, scale=TextLineCfg.SCALE, max_scale=TextLineCfg.MAX_SCALE)
                        arr[i] = img
                        meta[nm] = (None, scale)
                    else:
                        meta[nm] = (None, 1.0)
                if k + 2 < len(jobs):
                    ahead[k + 2] = decode(k + 2)                          # two batches decode while this one is on the GPU
                net.ctx.detect_submit(images=arr, slot=k & 1)
                if pending is not None:
                    slot, mem = pending
                    for nm, recs in zip(mem, net.ctx.detect_collect(slot, mode=mode, line_capacity=1024)):
                        results[nm] = recs
                pending = (k & 1, members)
            if pending is not None:
                slot, mem = pending
                for nm, recs in zip(mem, net.ctx.detect_collect(slot, mode=mode, line_capacity=1024)):
                    results[nm] = recs
    finally:
        if own_pool:
            pool.shutdown()
        for s_ in shms:
            s_.close()
            s_.unlink()
    for nm in singles:
        img, scale = _load(nm)
        from ctpn_amd.lib.fast_rcnn.test import test_ctpn
        from ctpn_amd.lib.text_connector.detectors import TextDetector
        scores, boxes = test_ctpn(None, net, img)
        results[nm] = TextDetector().detect(boxes, scores[:, np.newaxis], img.shape[:2])
        meta[nm] = (img, scale)
    dt = time.time() - t0
    for nm in names:
        img, scale = meta[nm]
        if write_images:
            if img is None:
                img, scale = _load(nm)
            D.draw_boxes(img.copy(), nm, results[nm], scale, out_dir)
        else:
            base = os.path.basename(nm)
            B.write_result_file(os.path.join(out_dir, 'res_{}.txt'.format(base.split('.')[0])), results[nm], scale)
    log('Detection of {:d} images in {:d} batches took {:.3f}s ({:.1f} images/s)'.format(len(names), len(jobs) + len(singles), dt, len(names) / max(dt, 1e-9)))
    return results


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument('--input', default='data/demo', help='directory or glob of images')
    ap.add_argument('--out', default='data/results')
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--mode', default=None, choices=[None, 'H', 'O'])
    ap.add_argument('--synthetic', type=int, default=None, metavar='SEED')
    ap.add_argument('--no-images', action='store_true', help='write only res_<stem>.txt')
    ap.add_argument('--decode-threads', type=int, default=8, help='host threads decoding / resizing the next batch')
    ap.add_argument('--decode-procs', type=int, default=0, help='decode in this many worker PROCESSES (shared-memory batches) instead of threads')
    ap.add_argument('--decode', default='host', choices=['host', 'gpu'], help="gpu: JPEG decode + resize_im on the device (ctpn_decode_jpeg_batch)")
    ap.add_argument('--precision', default=None, choices=['split', 'fp32', 'fp16', 'bf16'],
                    help="arithmetic of the conv stack; default: cfg.TEST.PRECISION (text.yml: split, the parity-grade mode). bf16 is the "
                         "throughput choice (3.2 x split's rate, outside the 1e-3 / 1 px bar)")
    args = ap.parse_args(argv)
    yml = 'ctpn/text.yml' if os.path.exists('ctpn/text.yml') else os.path.join(os.path.dirname(os.path.abspath(__file__)), 'text.yml')
    cfg_from_file(yml)
    if args.precision:
        cfg.TEST.PRECISION = args.precision
    net = get_network("VGGnet_test")
    D.load_weights(net, args.synthetic)
    names = list_images(args.input)
    if not names:
        raise SystemExit('no images under ' + args.input)
    run(net, names, args.out, batch=args.batch, mode=args.mode, write_images=not args.no_images, decode_threads=args.decode_threads,
        decode_procs=args.decode_procs, decode=args.decode)
    net.close()


if __name__ == '__main__':
    main()
