"""ORACLE (test infrastructure only -- never imported by the product path).

CPU fp32 restatement of the reference inference graph, op by op:
    lib/networks/VGGnet_test.py:16-55            topology
    lib/networks/network.py:160-183              conv  = relu(bias_add(conv2d(x, W[3,3,Ci,Co], stride 1, 'SAME')))
    lib/networks/network.py:189-196              max_pool 2x2/2 'VALID'
    lib/networks/network.py:88-113               Bilstm: reshape (N*H, W, C); fw/bw LSTMCell(128); concat; @W[256,512]+b
    lib/networks/network.py:144-158              lstm_fc: @W[512,40]+b, @W[512,20]+b
    lib/networks/network.py:269-277, 332-337     spatial_reshape(2) -> softmax -> spatial_reshape(20)
    lib/fast_rcnn/test.py:7-31, config.py:200    blob = float32(im) - PIXEL_MEANS (BGR)

PARITY UNPINNED for this file: the arithmetic of these ops lives in tensorflow_gpu==1.3.0 (reference
requirements.txt:2), which is not vendored under /root/reference and not installable here, and the reference
has no tests or golden tensors for them. The TF-1.3 semantics restated here (HWIO
cross-correlation with zero 'SAME' padding, LSTMCell gate order i, j, f, o with forget_bias 1.0, zero state,
bw output re-reversed, softmax over (bg, fg) pairs) are cross-checked in tests/test_oracle.py against a
hand-rolled numpy loop implementation on tiny shapes and -- the BiLSTM, whose TF semantics are the easiest to misread --
against torch.nn.LSTM's own recurrence with the parameters re-laid-out (bilstm_torch_nn): two implementations that share
no code with lstm_direction.
"""
import numpy as np
import torch
import torch.nn.functional as F

PIXEL_MEANS = np.array([[[102.9801, 115.9465, 122.7717]]])  # lib/fast_rcnn/config.py:200

CONVS = ["conv1_1", "conv1_2", "conv2_1", "conv2_2", "conv3_1", "conv3_2", "conv3_3", "conv4_1", "conv4_2",
         "conv4_3", "conv5_1", "conv5_2", "conv5_3", "rpn_conv/3x3"]
POOL_AFTER = {"conv1_2": "pool1", "conv2_2": "pool2", "conv3_3": "pool3", "conv4_3": "pool4"}


def image_blob(images_u8):
    """(n,h,w,3) uint8 BGR -> float32 NHWC, exactly numpy's in-place float32 -= float64 (test.py:8-9)."""
    im = np.asarray(images_u8)
    if im.ndim == 3:
        im = im[None]
    out = im.astype(np.float32, copy=True)
    out -= PIXEL_MEANS
    return out


def conv3x3_relu(x_nhwc, w_hwio, b, relu=True):
    x = torch.from_numpy(np.ascontiguousarray(x_nhwc)).permute(0, 3, 1, 2)
    w = torch.from_numpy(np.ascontiguousarray(w_hwio)).permute(3, 2, 0, 1)
    y = F.conv2d(x, w, torch.from_numpy(np.ascontiguousarray(b)), stride=1, padding=1)
    if relu:
        y = torch.relu(y)
    return y.permute(0, 2, 3, 1).contiguous().numpy()


def maxpool2x2(x_nhwc):
    x = torch.from_numpy(np.ascontiguousarray(x_nhwc)).permute(0, 3, 1, 2)
    y = F.max_pool2d(x, kernel_size=2, stride=2, padding=0)  # 'VALID': trailing odd row/col dropped
    return y.permute(0, 2, 3, 1).contiguous().numpy()


def _sigmoid(x):
    return 1.0 / (1.0 + torch.exp(-x))


def lstm_direction(x_rtc, kernel, bias, reverse):
    """TF-1.3 LSTMCell(128) over (rows, T, 512); returns (rows, T, 128)."""
    x = torch.from_numpy(np.ascontiguousarray(x_rtc))
    k = torch.from_numpy(np.ascontiguousarray(kernel))
    b = torch.from_numpy(np.ascontiguousarray(bias))
    R, T, _ = x.shape
    h = torch.zeros(R, 128)
    c = torch.zeros(R, 128)
    out = torch.zeros(R, T, 128)
    steps = range(T - 1, -1, -1) if reverse else range(T)
    for t in steps:
        z = torch.cat([x[:, t, :], h], dim=1) @ k + b
        i, j, f, o = torch.split(z, 128, dim=1)
        c = _sigmoid(f + 1.0) * c + _sigmoid(i) * torch.tanh(j)
        h = _sigmoid(o) * torch.tanh(c)
        out[:, t, :] = h
    return out.numpy()


def lstm_pre(x_nhwc, w):
    """x @ kernel[:512] + bias for both directions: (n,hf,wf,1024), fw gates then bw gates."""
    n, hf, wf, c = x_nhwc.shape
    x = torch.from_numpy(np.ascontiguousarray(x_nhwc)).reshape(-1, c)
    outs = []
    for d in ("fw", "bw"):
        k = torch.from_numpy(w["lstm_o/bidirectional_rnn/%s/lstm_cell/kernel" % d][:512])
        b = torch.from_numpy(w["lstm_o/bidirectional_rnn/%s/lstm_cell/bias" % d])
        outs.append(x @ k + b)
    return torch.cat(outs, dim=1).reshape(n, hf, wf, 1024).numpy()


def bilstm(x_nhwc, w):
    """-> lstm_out (n,hf,wf,256) = concat(fw, bw)."""
    n, hf, wf, c = x_nhwc.shape
    x = np.ascontiguousarray(x_nhwc).reshape(n * hf, wf, c)
    fw = lstm_direction(x, w["lstm_o/bidirectional_rnn/fw/lstm_cell/kernel"], w["lstm_o/bidirectional_rnn/fw/lstm_cell/bias"], False)
    bw = lstm_direction(x, w["lstm_o/bidirectional_rnn/bw/lstm_cell/kernel"], w["lstm_o/bidirectional_rnn/bw/lstm_cell/bias"], True)
    return np.concatenate([fw, bw], axis=-1).reshape(n, hf, wf, 256)


def bilstm_from_pre(pre_nhwc, w):
    """The RECURRENCE alone, from given pre-activations (n,hf,wf,1024) = x @ kernel[:512] + bias (fw gates | bw gates, TF order
    i, j, f, o): z_t = pre_t + h_{t-1} @ kernel[512:]. Pins the recurrent kernel at long T on the device's own lstm_pre (the
    input projection's bf16 operand rounding stays out of the comparison). -> lstm_out (n,hf,wf,256)."""
    n, hf, wf, c = pre_nhwc.shape
    assert c == 1024
    pre = torch.from_numpy(np.ascontiguousarray(pre_nhwc)).reshape(n * hf, wf, 1024)
    outs = []
    for d, (name, reverse) in enumerate((("fw", False), ("bw", True))):
        kh = torch.from_numpy(np.ascontiguousarray(w["lstm_o/bidirectional_rnn/%s/lstm_cell/kernel" % name][512:]))
        R = n * hf
        h = torch.zeros(R, 128)
        cst = torch.zeros(R, 128)
        out = torch.zeros(R, wf, 128)
        for t in (range(wf - 1, -1, -1) if reverse else range(wf)):
            z = pre[:, t, d * 512:(d + 1) * 512] + h @ kh
            i, j, f, o = torch.split(z, 128, dim=1)
            cst = _sigmoid(f + 1.0) * cst + _sigmoid(i) * torch.tanh(j)
            h = _sigmoid(o) * torch.tanh(cst)
            out[:, t, :] = h
        outs.append(out)
    return torch.cat(outs, dim=-1).reshape(n, hf, wf, 256).numpy()


def bilstm_torch_nn(x_nhwc, w):
    """INDEPENDENT second implementation of the BiLSTM (tests/test_oracle.py pins `bilstm` against it): torch.nn.LSTM's own
    recurrence kernel, fed the TF-1.3 LSTMCell parameters re-laid-out -- TF gate columns (i, j, f, o) -> torch rows (i, f, g, o),
    forget_bias 1.0 (tf.contrib.rnn.LSTMCell default, network.py:97) folded into the f bias, kernel[:512] / kernel[512:] ->
    weight_ih / weight_hh; bidirectional=True supplies the reversed pass and the re-reversal of its outputs
    (bidirectional_dynamic_rnn, network.py:100). Nothing of lstm_direction above is shared."""
    n, hf, wf, c = x_nhwc.shape
    lstm = torch.nn.LSTM(input_size=512, hidden_size=128, num_layers=1, batch_first=True, bidirectional=True)
    perm = np.concatenate([np.arange(0, 128), np.arange(256, 384), np.arange(128, 256), np.arange(384, 512)])   # i, f, j(=g), o
    with torch.no_grad():
        for name, sfx in (("fw", ""), ("bw", "_reverse")):
            k = np.asarray(w["lstm_o/bidirectional_rnn/%s/lstm_cell/kernel" % name], np.float32)
            b = np.asarray(w["lstm_o/bidirectional_rnn/%s/lstm_cell/bias" % name], np.float32).copy()
            b[256:384] += 1.0
            getattr(lstm, "weight_ih_l0" + sfx).copy_(torch.from_numpy(np.ascontiguousarray(k[:512][:, perm].T)))
            getattr(lstm, "weight_hh_l0" + sfx).copy_(torch.from_numpy(np.ascontiguousarray(k[512:][:, perm].T)))
            getattr(lstm, "bias_ih_l0" + sfx).copy_(torch.from_numpy(b[perm]))
            getattr(lstm, "bias_hh_l0" + sfx).zero_()
        y, _ = lstm(torch.from_numpy(np.ascontiguousarray(x_nhwc)).reshape(n * hf, wf, c))
    return y.reshape(n, hf, wf, 256).numpy()


def dense(x_nhwc, wmat, b):
    shp = x_nhwc.shape
    x = torch.from_numpy(np.ascontiguousarray(x_nhwc)).reshape(-1, shp[-1])
    y = x @ torch.from_numpy(np.ascontiguousarray(wmat)) + torch.from_numpy(np.ascontiguousarray(b))
    return y.reshape(shp[:-1] + (wmat.shape[1],)).numpy()


def pair_softmax(cls_score):
    """(n,h,w,20) -> softmax over (bg,fg) pairs, anchor-major / class-minor channel layout."""
    n, h, w, c = cls_score.shape
    s = torch.from_numpy(np.ascontiguousarray(cls_score)).reshape(n, h, w * (c // 2), 2)
    p = torch.softmax(s, dim=-1)
    return p.reshape(n, h, w, c).numpy()


def forward(images_u8, weights, keep=None, blob=None):
    """Full forward. weights: dict name -> array (ctpn_amd.arena_views). Returns dict of NHWC fp32 arrays;
    `keep` limits which intermediate names are retained (None = all). blob: feed this (n,h,w,3) float32 net.data blob
    (test.py:47-49, after _get_image_blob's rescale) instead of uint8 images."""
    torch.set_grad_enabled(False)
    out = {}

    def put(name, v):
        if keep is None or name in keep:
            out[name] = v

    x = image_blob(images_u8) if blob is None else np.ascontiguousarray(blob, dtype=np.float32)
    for name in CONVS:
        x = conv3x3_relu(x, weights[name + "/weights"], weights[name + "/biases"])
        put(name, x)
        if name in POOL_AFTER:
            x = maxpool2x2(x)
            put(POOL_AFTER[name], x)
    put("lstm_pre", lstm_pre(x, weights)) if (keep is None or "lstm_pre" in keep) else None
    lo = bilstm(x, weights)
    put("lstm_out", lo)
    fc = dense(lo, weights["lstm_o/weights"], weights["lstm_o/biases"])
    put("lstm_o", fc)
    bbox = dense(fc, weights["rpn_bbox_pred/weights"], weights["rpn_bbox_pred/biases"])
    cls = dense(fc, weights["rpn_cls_score/weights"], weights["rpn_cls_score/biases"])
    put("rpn_cls_score", cls)
    out["rpn_bbox_pred"] = bbox
    out["rpn_cls_prob_reshape"] = pair_softmax(cls)
    return out
