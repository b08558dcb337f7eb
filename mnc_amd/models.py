"""Programmatic description of the VGG-16 MNC 5-stage INFERENCE graph.

The engine executes any prototxt given to `caffe.Net` (e.g. the reference's own
models/VGG16/mnc_5stage/test.prototxt passed with --def).  For tests, the bench and the GPU box -- where the
reference checkout is not available -- this module EMITS an equivalent prototxt from a compact builder, so no model
file has to be vendored.  tests/test_graph_equivalence.py checks layer by layer (names, types, bottoms/tops, shared
parameter names, hyper-parameters) that the emitted graph equals the reference file when that is mounted.

Graph facts encoded here (citations into the reference's test.prototxt):
  trunk conv1_1..conv5_3 with 4 MAX 2x2/2 pools (:19-387); RPN head + 2-way softmax via Reshape (:391-462);
  `proposal` Python layer (:463-475); stage 2 warps 28x28 then pools to 14x14 (:479-505) while stage 4 warps
  directly to 14x14 (:809-820); Concat order is (fc7_mask, fc7) (:700-709); stage 4/5 share all 9 weight/bias pairs
  with stage 2/3 through `param { name }` (:514-515 <-> :829-834, ...)."""
import os
import tempfile

VGG_CFG = [("1_1", 64), ("1_2", 64), "P1", ("2_1", 128), ("2_2", 128), "P2", ("3_1", 256), ("3_2", 256), ("3_3", 256),
           "P3", ("4_1", 512), ("4_2", 512), ("4_3", 512), "P4", ("5_1", 512), ("5_2", 512), ("5_3", 512)]


class _Emit(object):
    def __init__(self, name):
        self.lines = ['name: "%s"' % name]

    def raw(self, text):
        self.lines.append(text)

    def layer(self, name, typ, bottoms, tops, body="", params=None):
        out = ["layer {", '  name: "%s"' % name, '  type: "%s"' % typ]
        out += ['  bottom: "%s"' % b for b in bottoms]
        out += ['  top: "%s"' % t for t in tops]
        for p in params or []:
            out.append('  param { name: "%s" }' % p)
        if body:
            out.append("  " + body)
        out.append("}")
        self.lines.append("\n".join(out))

    def text(self):
        return "\n".join(self.lines) + "\n"


def _conv(e, name, bottom, top, n_out, k, pad, stride=1, bias=True):
    e.layer(name, "Convolution", [bottom], [top],
            "convolution_param { num_output: %d kernel_size: %d pad: %d stride: %d%s }"
            % (n_out, k, pad, stride, "" if bias else " bias_term: false"))


def _relu(e, name, blob):
    e.layer(name, "ReLU", [blob], [blob])


def _pool(e, name, bottom, top):
    e.layer(name, "Pooling", [bottom], [top], "pooling_param { pool: MAX kernel_size: 2 stride: 2 pad: 0 }")


def _fc(e, name, bottom, top, n_out, pname=None):
    e.layer(name, "InnerProduct", [bottom], [top], "inner_product_param { num_output: %d }" % n_out,
            params=[pname + "_w", pname + "_b"] if pname else None)


def _head(e, sfx, rois, warp_direct, d=1, trunk_top="conv5_3"):
    """Stages 2+3 (sfx '') or 4+5 (sfx '_ext'): mask estimation + box/mask classification on `rois`."""
    wide, narrow = 4096 // d, max(256 // d, 32)
    feat = "roi_interpolate_conv5" + sfx
    if warp_direct:
        e.layer(feat, "ROIWarping", [trunk_top, rois], [feat],
                "roi_warping_param { pooled_w: 14 pooled_h: 14 spatial_scale: 0.0625 }")
    else:
        pre = "roi_interpolate_conv5_premax" + sfx
        e.layer(pre, "ROIWarping", [trunk_top, rois], [pre],
                "roi_warping_param { pooled_w: 28 pooled_h: 28 spatial_scale: 0.0625 }")
        _pool(e, feat, pre, feat)
    _fc(e, "fc6_maskest" + sfx, feat, "fc6_maskest" + sfx, narrow, "fc6_maskest")
    _relu(e, "relu6_maskest" + sfx, "fc6_maskest" + sfx)
    _fc(e, "mask_pred" + sfx, "fc6_maskest" + sfx, "mask_pred" + sfx, 441, "mask_pred")
    e.layer("mask_output" + sfx, "Sigmoid", ["mask_pred" + sfx], ["mask_output" + sfx])
    e.layer("mask_proposal" + sfx, "Python", ["mask_output" + sfx], ["mask_proposal" + sfx],
            "python_param { module: 'pylayer.mask_layer' layer: 'MaskLayer' }")
    e.layer("mask_resize" + sfx, "MaskResize", ["mask_proposal" + sfx], ["mask_proposal_resize" + sfx],
            "mask_resize_param { output_height: 14 output_width: 14 }")
    _pool(e, "roi_interpolate_conv5_box" + sfx, feat, "roi_interpolate_conv5_box" + sfx)
    _fc(e, "fc6" + sfx, "roi_interpolate_conv5_box" + sfx, "fc6" + sfx, wide, "fc6")
    _relu(e, "relu6" + sfx, "fc6" + sfx)
    _fc(e, "fc7" + sfx, "fc6" + sfx, "fc7" + sfx, wide, "fc7")
    _relu(e, "relu7" + sfx, "fc7" + sfx)
    e.layer("mask_pooling" + sfx, "MaskPooling", [feat, "mask_proposal_resize" + sfx], ["roi_mask_conv5" + sfx])
    _pool(e, "roi_interpolate_conv5_mask" + sfx, "roi_mask_conv5" + sfx, "roi_interpolate_conv5_mask" + sfx)
    _fc(e, "fc6_mask" + sfx, "roi_interpolate_conv5_mask" + sfx, "fc6_mask" + sfx, wide, "fc6_mask")
    _relu(e, "relu6_mask" + sfx, "fc6_mask" + sfx)
    _fc(e, "fc7_mask" + sfx, "fc6_mask" + sfx, "fc7_mask" + sfx, wide, "fc7_mask")
    _relu(e, "relu7_mask" + sfx, "fc7_mask" + sfx)
    e.layer("join_box_mask" + sfx, "Concat", ["fc7_mask" + sfx, "fc7" + sfx], ["join_box_mask" + sfx],
            "concat_param { axis: 1 }")
    _fc(e, "cls_score" + sfx, "join_box_mask" + sfx, "cls_score" + sfx, 21, "cls_score")
    e.layer("cls_prob" + sfx, "Softmax", ["cls_score" + sfx], ["cls_prob" + sfx])
    _fc(e, "seg_cls_score" + sfx, "join_box_mask" + sfx, "seg_cls_score" + sfx, 21, "seg_cls_score")
    e.layer("seg_cls_prob" + sfx, "Softmax", ["seg_cls_score" + sfx], ["seg_cls_prob" + sfx])
    _fc(e, "bbox_pred" + sfx, "join_box_mask" + sfx, "bbox_pred" + sfx, 84, "bbox_pred")


def _trunk(e, d):
    """conv1_1 .. conv5_3 with the four MAX 2x2/2 pools (every reference test graph, e.g. mnc_5stage/test.prototxt:19-387)."""
    bottom = "data"
    for item in VGG_CFG:
        if isinstance(item, str):
            top = "pool" + item[1]
            _pool(e, top, bottom, top)
        else:
            tag, width = item
            top = "conv" + tag
            _conv(e, top, bottom, top, max(width // d, 32), 3, 1)
            _relu(e, "relu" + tag, top)
        bottom = top


def _trunk_rpn_proposal(width_div):
    """VGG-16 trunk, RPN head and the ProposalLayer -- shared by the two RPN test graphs of the reference
    (models/VGG16/{mnc_5stage,faster_rcnn_end2end}/test.prototxt:1-470)."""
    d = width_div
    e = _Emit("VGG16")
    e.raw('input: "data"\ninput_shape { dim: 1 dim: 3 dim: 224 dim: 224 }')
    e.raw('input: "im_info"\ninput_shape { dim: 1 dim: 3 }')
    _trunk(e, d)
    _rpn_proposal(e, d, "conv5_3")
    return e, d


def _rpn_proposal(e, d, trunk_top):
    """RPN head on the trunk's top (test.prototxt:391-475): 3x3 conv + ReLU, the two 1x1 heads, 2-way softmax via Reshape,
    ProposalLayer."""
    _conv(e, "rpn_conv_3x3", trunk_top, "rpn_output", max(512 // d, 32), 3, 1)
    _relu(e, "rpn_relu_3x3", "rpn_output")
    _conv(e, "rpn_cls_score", "rpn_output", "rpn_cls_score", 18, 1, 0)
    _conv(e, "rpn_bbox_pred", "rpn_output", "rpn_bbox_pred", 36, 1, 0)
    e.layer("rpn_cls_score_reshape", "Reshape", ["rpn_cls_score"], ["rpn_cls_score_reshape"],
            "reshape_param { shape { dim: 0 dim: 2 dim: -1 dim: 0 } }")
    e.layer("rpn_cls_prob", "Softmax", ["rpn_cls_score_reshape"], ["rpn_cls_prob"])
    e.layer("rpn_cls_prob_reshape", "Reshape", ["rpn_cls_prob"], ["rpn_cls_prob_reshape"],
            "reshape_param { shape { dim: 0 dim: 18 dim: -1 dim: 0 } }")
    e.layer("proposal", "Python", ["rpn_cls_prob_reshape", "rpn_bbox_pred", "im_info"], ["rois"],
            "python_param { module: 'pylayer.proposal_layer' layer: 'ProposalLayer' "
            "param_str: \"{'feat_stride': 16, 'gradient_scale': 1}\" }")


def mnc_5stage_test_prototxt(width_div=1):
    """Text of the 5-stage test graph.  width_div > 1 divides every trunk/FC width (a reduced net for quick executor
    tests; 1 is the real VGG-16 model)."""
    e, d = _trunk_rpn_proposal(width_div)
    _head(e, "", "rois", False, d)
    e.layer("stage_bridge", "Python", ["rois", "bbox_pred", "seg_cls_prob", "im_info"], ["rois_ext"],
            "python_param { module: 'pylayer.stage_bridge_layer' layer: 'StageBridgeLayer' }")
    _head(e, "_ext", "rois_ext", True, d)
    return e.text()


def faster_rcnn_end2end_test_prototxt(width_div=1):
    """Text of the Faster R-CNN end2end test graph (models/VGG16/faster_rcnn_end2end/test.prototxt:479-620): the same
    trunk + RPN + ProposalLayer, then ROIWarping 7x7 -> fc6 -> fc7 (ReLU + test-time-identity Dropout) -> cls_score /
    bbox_pred -> cls_prob.  SURVEY section 8f row n3: it runs on the kernels of the MNC path unchanged."""
    e, d = _trunk_rpn_proposal(width_div)
    wide = 4096 // d
    e.layer("roi_pool5", "ROIWarping", ["conv5_3", "rois"], ["pool5"],
            "roi_warping_param { pooled_w: 7 pooled_h: 7 spatial_scale: 0.0625 }")
    _fc(e, "fc6", "pool5", "fc6", wide)
    _relu(e, "relu6", "fc6")
    e.layer("drop6", "Dropout", ["fc6"], ["fc6"], "dropout_param { dropout_ratio: 0.5 }")
    _fc(e, "fc7", "fc6", "fc7", wide)
    _relu(e, "relu7", "fc7")
    e.layer("drop7", "Dropout", ["fc7"], ["fc7"], "dropout_param { dropout_ratio: 0.5 }")
    _fc(e, "cls_score", "fc7", "cls_score", 21)
    _fc(e, "bbox_pred", "fc7", "bbox_pred", 84)
    e.layer("cls_prob", "Softmax", ["cls_score"], ["cls_prob"])
    return e.text()


def cfm_test_prototxt(width_div=1):
    """Text of the CFM (convolutional feature masking) test graph, models/VGG16/cfm/test.prototxt: no RPN -- `rois`
    (batch index + box per MCG proposal, over a batch of pyramid levels) and binary `masks` are inputs; ROIPooling 7x7 ->
    fc6/fc7 (box feature), ROIPooling 14x14 -> MaskPooling -> pool -> fc6_mask/fc7_mask (mask feature) and -> fc6_maskest ->
    mask_pred -> mask_prob; Concat (fc7_mask, fc7) -> cls / seg_cls / bbox heads (:395-620).  SURVEY section 8f row n3."""
    d = width_div
    e = _Emit("VGG16")
    e.raw('input: "data"\ninput_shape { dim: 1 dim: 3 dim: 224 dim: 224 }')
    e.raw('input: "rois"\ninput_shape { dim: 1 dim: 5 }')
    e.raw('input: "masks"\ninput_shape { dim: 1 dim: 1 dim: 14 dim: 14 }')
    _trunk(e, d)
    wide, narrow = 4096 // d, max(256 // d, 32)
    e.layer("roi_pooling_conv5", "ROIPooling", ["conv5_3", "rois"], ["roi_pooling_conv5"],
            "roi_pooling_param { pooled_w: 7 pooled_h: 7 spatial_scale: 0.0625 }")
    _fc(e, "fc6", "roi_pooling_conv5", "fc6", wide)
    _relu(e, "relu6", "fc6")
    _fc(e, "fc7", "fc6", "fc7", wide)
    _relu(e, "relu7", "fc7")
    e.layer("roi_pooling_conv5_mask", "ROIPooling", ["conv5_3", "rois"], ["roi_pooling_conv5_mask"],
            "roi_pooling_param { pooled_w: 14 pooled_h: 14 spatial_scale: 0.0625 }")
    e.layer("mask_pooling", "MaskPooling", ["roi_pooling_conv5_mask", "masks"], ["roi_mask_conv5"])
    _pool(e, "roi_mask_conv5", "roi_mask_conv5", "roi_mask_conv5_pool")
    _fc(e, "fc6_mask", "roi_mask_conv5_pool", "fc6_mask", wide)
    _relu(e, "relu6_mask", "fc6_mask")
    _fc(e, "fc7_mask", "fc6_mask", "fc7_mask", wide)
    _relu(e, "relu7_mask", "fc7_mask")
    _fc(e, "fc6_maskest", "roi_pooling_conv5_mask", "fc6_maskest", narrow)
    _relu(e, "relu6_maskest", "fc6_maskest")
    _fc(e, "mask_pred", "fc6_maskest", "mask_pred", 441)
    e.layer("mask_prob", "Sigmoid", ["mask_pred"], ["mask_prob"])
    e.layer("join_box_mask", "Concat", ["fc7_mask", "fc7"], ["join_box_mask"], "concat_param { axis: 1 }")
    _fc(e, "cls_score", "join_box_mask", "cls_score", 21)
    e.layer("cls_prob", "Softmax", ["cls_score"], ["cls_prob"])
    _fc(e, "seg_cls_score", "join_box_mask", "seg_cls_score", 21)
    e.layer("seg_cls_prob", "Softmax", ["seg_cls_score"], ["seg_cls_prob"])
    _fc(e, "bbox_pred", "join_box_mask", "bbox_pred", 84)
    return e.text()


def write_cfm_test_prototxt(path=None, width_div=1):
    return _write(cfm_test_prototxt(width_div), path, "cfm_test_w%d.prototxt" % width_div)


RESNET50_STAGES = [(2, 3, 64, 1), (3, 4, 128, 2), (4, 6, 256, 2)]          # (stage, blocks, bottleneck width, first stride): C4


def _bn_scale_relu(e, tag, blob, relu=True):
    """conv -> BatchNorm(use_global_stats) -> Scale(bias) [-> ReLU], all in place: the ResNet deploy idiom."""
    e.layer("bn" + tag, "BatchNorm", [blob], [blob], "batch_norm_param { use_global_stats: true }")
    e.layer("scale" + tag, "Scale", [blob], [blob], "scale_param { bias_term: true }")
    if relu:
        _relu(e, blob + "_relu", blob)


def _resnet50_c4(e, d):
    """ResNet-50 conv1 .. res4f in the layout of the public ResNet-50-deploy.prototxt (He et al.): 7x7/2 stem + BN/Scale/ReLU,
    MAX 3x3/2, bottleneck blocks res{2,3,4}{a..} with the stride on branch1 / branch2a of each stage's first block, no conv
    biases, Eltwise SUM + ReLU.  Output `res4f`: 1024 // d channels at stride 16 -- the role conv5_3 plays for VGG-16."""
    _conv(e, "conv1", "data", "conv1", max(64 // d, 16), 7, 3, stride=2)
    _bn_scale_relu(e, "_conv1", "conv1")
    e.layer("pool1", "Pooling", ["conv1"], ["pool1"], "pooling_param { pool: MAX kernel_size: 3 stride: 2 }")
    prev = "pool1"
    for stage, blocks, width, first_stride in RESNET50_STAGES:
        mid, wide = max(width // d, 8), max(4 * width // d, 32)
        for bi in range(blocks):
            tag = "%d%s" % (stage, "abcdef"[bi])
            stride = first_stride if bi == 0 else 1
            if bi == 0:
                _conv(e, "res%s_branch1" % tag, prev, "res%s_branch1" % tag, wide, 1, 0, stride, bias=False)
                _bn_scale_relu(e, "%s_branch1" % tag, "res%s_branch1" % tag, relu=False)
                shortcut = "res%s_branch1" % tag
            else:
                shortcut = prev
            a, b, c = ("res%s_branch2%s" % (tag, x) for x in "abc")
            _conv(e, a, prev, a, mid, 1, 0, stride, bias=False)
            _bn_scale_relu(e, "%s_branch2a" % tag, a)
            _conv(e, b, a, b, mid, 3, 1, 1, bias=False)
            _bn_scale_relu(e, "%s_branch2b" % tag, b)
            _conv(e, c, b, c, wide, 1, 0, 1, bias=False)
            _bn_scale_relu(e, "%s_branch2c" % tag, c, relu=False)
            e.layer("res" + tag, "Eltwise", [shortcut, c], ["res" + tag])
            _relu(e, "res%s_relu" % tag, "res" + tag)
            prev = "res" + tag
    return prev


def mnc_resnet50_test_prototxt(width_div=1):
    """The 5-stage MNC test graph on a ResNet-50 C4 trunk (BASELINE.json configs[4]).  The reference
    ships only VGG-16 models: this graph is this project's own composition -- the public ResNet-50 deploy trunk up to res4f
    (stride 16, 1024 channels) in place of conv1_1..conv5_3, then the reference's RPN and cascade heads unchanged
    (`_rpn_proposal`, `_head`: same layer names, shared-parameter names and Python layers as mnc_5stage/test.prototxt)."""
    d = width_div
    e = _Emit("ResNet50")
    e.raw('input: "data"\ninput_shape { dim: 1 dim: 3 dim: 224 dim: 224 }')
    e.raw('input: "im_info"\ninput_shape { dim: 1 dim: 3 }')
    top = _resnet50_c4(e, d)
    _rpn_proposal(e, d, top)
    _head(e, "", "rois", False, d, trunk_top=top)
    e.layer("stage_bridge", "Python", ["rois", "bbox_pred", "seg_cls_prob", "im_info"], ["rois_ext"],
            "python_param { module: 'pylayer.stage_bridge_layer' layer: 'StageBridgeLayer' }")
    _head(e, "_ext", "rois_ext", True, d, trunk_top=top)
    return e.text()


def write_mnc_resnet50_test_prototxt(path=None, width_div=1):
    return _write(mnc_resnet50_test_prototxt(width_div), path, "mnc_resnet50_test_w%d.prototxt" % width_div)


def write_faster_rcnn_end2end_test_prototxt(path=None, width_div=1):
    return _write(faster_rcnn_end2end_test_prototxt(width_div), path, "faster_rcnn_end2end_test_w%d.prototxt" % width_div)


def write_mnc_5stage_test_prototxt(path=None, width_div=1):
    """Write the graph to `path` (default: a per-user temp file) and return the path."""
    return _write(mnc_5stage_test_prototxt(width_div), path, "mnc_5stage_test_w%d.prototxt" % width_div)


def _write(text, path, default_name):
    if path is None:
        d = os.path.join(tempfile.gettempdir(), "mnc_amd_models")
        os.makedirs(d, exist_ok=True)
        path = os.path.join(d, default_name)
    tmp = path + ".%d.tmp" % os.getpid()
    with open(tmp, "w") as f:
        f.write(text)
    os.replace(tmp, path)
    return path
