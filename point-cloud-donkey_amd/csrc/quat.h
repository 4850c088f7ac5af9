// quat.h — Utils::getRotQuaternion / matrix2Quat / quatRotate / quatRotateInv (utils/utils.cpp:136-178, 342-394, 560-574) on the
// device, shared by vote casting (codebook.hip) and training (train.hip). Included inside an anonymous namespace.
#pragma once
struct Quat { float w, x, y, z; };
__device__ __forceinline__ Quat qmul(const Quat& a, const Quat& b) {     // boost::math::quaternion operator*
    Quat r;
    r.w = +a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
    r.x = +a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
    r.y = +a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x;
    r.z = +a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w;
    return r;
}
__device__ __forceinline__ Quat qconj(const Quat& q) { return Quat{q.w, -q.x, -q.y, -q.z}; }

// Utils::getRotQuaternion + matrix2Quat: rows of the matrix are the LRF axes
__device__ __forceinline__ Quat rot_quaternion(const float* l) {
    const float m[3][3] = {{l[0], l[1], l[2]}, {l[3], l[4], l[5]}, {l[6], l[7], l[8]}};
    float q[4] = {0.f, 0.f, 0.f, 1.f};   // x,y,z,w
    const float trace = m[0][0] + m[1][1] + m[2][2];
    float root;
    if (trace > 0.0f) {
        root = sqrtf(trace + 1.0f);
        q[3] = 0.5f * root;
        root = 0.5f / root;
        q[0] = (m[2][1] - m[1][2]) * root;
        q[1] = (m[0][2] - m[2][0]) * root;
        q[2] = (m[1][0] - m[0][1]) * root;
    } else {
        int i = 0;
        if (m[1][1] > m[0][0]) i = 1;
        if (m[2][2] > m[i][i]) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        root = sqrtf((float)((double)(m[i][i] - m[j][j] - m[k][k]) + 1.0));
        q[i] = 0.5f * root;
        root = 0.5f / root;
        q[3] = (m[k][j] - m[j][k]) * root;
        q[j] = (m[j][i] + m[i][j]) * root;
        q[k] = (m[k][i] + m[i][k]) * root;
    }
    return Quat{q[3], q[0], q[1], q[2]};
}

