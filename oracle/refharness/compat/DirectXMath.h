// The reference's Core/Math.cpp:2,111-114 uses DirectXMath only for XMMatrixMultiply (row-major
// 4x4 product).  Minimal equivalent; per element ((a0*b0 + a1*b1) + a2*b2) + a3*b3.
#pragma once
namespace DirectX {
struct XMFLOAT4X4 { float _11, _12, _13, _14, _21, _22, _23, _24, _31, _32, _33, _34, _41, _42, _43, _44; };
struct XMMATRIX {
    float m[4][4];
    XMMATRIX() {}
    explicit XMMATRIX(const float* p) { for (int i = 0; i < 16; ++i) (&m[0][0])[i] = p[i]; }
};
inline XMMATRIX XMMatrixMultiply(const XMMATRIX& a, const XMMATRIX& b) {
    XMMATRIX r;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j)
            r.m[i][j] = ((a.m[i][0] * b.m[0][j] + a.m[i][1] * b.m[1][j]) + a.m[i][2] * b.m[2][j]) + a.m[i][3] * b.m[3][j];
    return r;
}
inline void XMStoreFloat4x4(XMFLOAT4X4* d, const XMMATRIX& s) { for (int i = 0; i < 16; ++i) (&d->_11)[i] = (&s.m[0][0])[i]; }
}
