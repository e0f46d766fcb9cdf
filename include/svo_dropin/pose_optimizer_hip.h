// pose_optimizer_hip.h -- declaration of the batched structure refinement defined in pose_optimizer_hip.cpp
// (pose_optimizer::optimizeGaussNewton itself keeps the reference's own header, I/pose_optimizer.h).
#ifndef SVO_POSE_OPTIMIZER_HIP_H_
#define SVO_POSE_OPTIMIZER_HIP_H_

#include <svo/global.h>

namespace svo {

/// FrameHandlerBase::optimizeStructure (frame_handler_base.cpp:190-210) with the Point::optimize calls batched into
/// one device launch; same point selection and bookkeeping.
void optimizeStructureHip(FramePtr frame, size_t max_n_pts, int max_iter);

}  // namespace svo

#endif  // SVO_POSE_OPTIMIZER_HIP_H_
