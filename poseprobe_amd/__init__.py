"""poseprobe_amd - MI355X-native implementation of PoseProbe's object-branch hot path (voxel-grid volume renderer
with joint SE(3) pose optimisation).  See DESIGN.md / INTEGRATION.md."""
__version__ = '0.1.0'
