// The exact camera matrix for the drop-in PT::Pathtracer, without touching the reference's headers.
//
// Camera::generate_ray (student/camera.cpp:7-34) transforms its ray with the private member `iview`, which update_pos
// builds as Mat4::translate(position) * rot.to_mat() (util/camera.cpp:141); `rot` and `iview` have no getter, and the public
// get_view().inverse() differs from iview in the last bits (it is the inverse of an inverse).  An explicit template
// instantiation may name a private member - access checking does not apply to explicit instantiations ([temp.spec]) - so
// the cached matrix itself can be read: same bits, no patch.  tests/test_dropin_cpu.py proves it against the
// reference's Camera after look_at / orbit / move / zoom sequences.
#pragma once

#include "../util/camera.h"

namespace srt_host {

template <typename Tag, typename Tag::type Member> struct PrivateMember {
    friend typename Tag::type srt_private_member(Tag) { return Member; }
};
struct CameraIviewTag {
    typedef Mat4 Camera::*type;
    friend type srt_private_member(CameraIviewTag);
};
template struct PrivateMember<CameraIviewTag, &Camera::iview>;

inline const Mat4& camera_iview(const Camera& cam) {
    return cam.*srt_private_member(CameraIviewTag());
}

}  // namespace srt_host
