"""Counterpart of the reference's ``util/generateFeature.py``: joint uvd <-> dense offset/heat maps,
as fused HIP kernels (forward and backward) instead of ~20 torch ops with (B,J,3,S,S) temporaries."""
from .. import ops


class GFM:
    channels_last = True      # maps come out with channels-last strides (same values), matching the convolution outputs

    def joint2offset(self, joint, img, kernel_size, feature_size):
        """joint (B,J,3) crop-normalised uvd, img (B,1,H,H) -> (B,4J,S,S) unit offsets + heat
        (/root/reference/util/generateFeature.py:14-37)."""
        return ops.Joint2Offset.apply(joint, img, float(kernel_size), int(feature_size), self.channels_last)

    def offset2joint_softmax(self, offset, depth, kernel_size, scale=30):
        """(B,4J,S,S) maps + depth -> (B,J,3) by softmax(scale*heat)-weighted voting (reference :39-59)."""
        return ops.Offset2Joint.apply(offset, depth, float(kernel_size), float(scale))

    def feature2joint(self, img, pixel_pd, feature_types, feature_paras):
        joint = None
        for i, ft in enumerate(feature_types):
            if ft == 'offset':
                joint = self.offset2joint_softmax(pixel_pd, img, feature_paras[i])
        return joint

    def joint2feature(self, joint, img, feature_paras, feature_size, feature_types):
        feature = None
        for i, ft in enumerate(feature_types):
            if ft == 'offset':
                feature = self.joint2offset(joint, img, feature_paras[i], feature_size)
        return feature


def joint2offset(joint, img, kernel_size, feature_size):
    """module-level twin used by the backbone's stage-2 remap (/root/reference/model/backbone.py:68-91)."""
    return ops.Joint2Offset.apply(joint, img, float(kernel_size), int(feature_size), True)


def offset2joint_softmax(offset, depth, kernel_size, scale=30):
    return ops.Offset2Joint.apply(offset, depth, float(kernel_size), float(scale))
