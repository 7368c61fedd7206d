// hpmvs/Vec.h -- the fixed-size vector subset of Eigen that the reference's PUBLIC interface of the
// refinement path exposes (Patch3d::center_/normal_/color_, Camera axes: reference
// include/hpmvs/Patch3d.h:55-56,73, Camera.h:87-105).  If the real Eigen3 is installed it is used
// instead (same type names), so callers written against the reference headers compile either way.
#ifndef HPMVS_VEC_H_
#define HPMVS_VEC_H_

#if defined(__has_include)
#if __has_include(<Eigen/Dense>) && !defined(HPMVS_NO_EIGEN)
#define HPMVS_HAVE_EIGEN 1
#include <Eigen/Dense>
#endif
#endif

#ifndef HPMVS_HAVE_EIGEN
#include <cmath>
#include <cstddef>

namespace Eigen {

template <typename T, int N>
struct Matrix_ {  // column vector of N scalars
    T v[N];
    Matrix_() { for (int i = 0; i < N; i++) v[i] = T(0); }
    Matrix_(T a, T b) { static_assert(N == 2, "size"); v[0] = a; v[1] = b; }
    Matrix_(T a, T b, T c) { static_assert(N == 3, "size"); v[0] = a; v[1] = b; v[2] = c; }
    Matrix_(T a, T b, T c, T d) { static_assert(N == 4, "size"); v[0] = a; v[1] = b; v[2] = c; v[3] = d; }
    static Matrix_ Zero() { return Matrix_(); }
    T& operator[](int i) { return v[i]; }
    const T& operator[](int i) const { return v[i]; }
    T& operator()(int i) { return v[i]; }
    const T& operator()(int i) const { return v[i]; }
    T* data() { return v; }
    const T* data() const { return v; }
    static int size() { return N; }
    T squaredNorm() const { T s = T(0); for (int i = 0; i < N; i++) s += v[i] * v[i]; return s; }
    T norm() const { return std::sqrt(squaredNorm()); }
    T dot(const Matrix_& o) const { T s = T(0); for (int i = 0; i < N; i++) s += v[i] * o.v[i]; return s; }
    Matrix_ normalized() const { T n2 = squaredNorm(); Matrix_ r = *this; if (n2 > T(0)) { T n = std::sqrt(n2); for (int i = 0; i < N; i++) r.v[i] = v[i] / n; } return r; }
    void normalize() { *this = normalized(); }
    Matrix_ cross(const Matrix_& o) const {
        static_assert(N == 3, "cross is 3-d");
        return Matrix_(v[1] * o.v[2] - v[2] * o.v[1], v[2] * o.v[0] - v[0] * o.v[2], v[0] * o.v[1] - v[1] * o.v[0]);
    }
    Matrix_ operator+(const Matrix_& o) const { Matrix_ r; for (int i = 0; i < N; i++) r.v[i] = v[i] + o.v[i]; return r; }
    Matrix_ operator-(const Matrix_& o) const { Matrix_ r; for (int i = 0; i < N; i++) r.v[i] = v[i] - o.v[i]; return r; }
    Matrix_ operator-() const { Matrix_ r; for (int i = 0; i < N; i++) r.v[i] = -v[i]; return r; }
    Matrix_ operator*(T s) const { Matrix_ r; for (int i = 0; i < N; i++) r.v[i] = v[i] * s; return r; }
    Matrix_ operator/(T s) const { Matrix_ r; for (int i = 0; i < N; i++) r.v[i] = v[i] / s; return r; }
    Matrix_& operator+=(const Matrix_& o) { for (int i = 0; i < N; i++) v[i] += o.v[i]; return *this; }
    Matrix_& operator-=(const Matrix_& o) { for (int i = 0; i < N; i++) v[i] -= o.v[i]; return *this; }
    Matrix_& operator*=(T s) { for (int i = 0; i < N; i++) v[i] *= s; return *this; }
    Matrix_& operator/=(T s) { for (int i = 0; i < N; i++) v[i] /= s; return *this; }
    bool operator==(const Matrix_& o) const { for (int i = 0; i < N; i++) if (v[i] != o.v[i]) return false; return true; }
    // head<3>() / head(3): an assignable view of the first M coefficients
    template <int M> struct Head {
        T* p;
        Head& operator=(const Matrix_<T, M>& o) { for (int i = 0; i < M; i++) p[i] = o.v[i]; return *this; }
        operator Matrix_<T, M>() const { Matrix_<T, M> r; for (int i = 0; i < M; i++) r.v[i] = p[i]; return r; }
        T norm() const { return Matrix_<T, M>(*this).norm(); }
        Matrix_<T, M> normalized() const { return Matrix_<T, M>(*this).normalized(); }
        T dot(const Matrix_<T, M>& o) const { return Matrix_<T, M>(*this).dot(o); }
        Matrix_<T, M> operator-(const Matrix_<T, M>& o) const { return Matrix_<T, M>(*this) - o; }
        Matrix_<T, M> operator+(const Matrix_<T, M>& o) const { return Matrix_<T, M>(*this) + o; }
        T operator[](int i) const { return p[i]; }
    };
    template <int M> Head<M> head() { return Head<M>{v}; }
    template <int M> Matrix_<T, M> head() const { Matrix_<T, M> r; for (int i = 0; i < M; i++) r.v[i] = v[i]; return r; }
    Head<3> head(int) { return Head<3>{v}; }
    template <typename U> Matrix_<U, N> cast() const { Matrix_<U, N> r; for (int i = 0; i < N; i++) r.v[i] = (U)v[i]; return r; }
};
template <typename T, int N> inline Matrix_<T, N> operator*(T s, const Matrix_<T, N>& a) { return a * s; }

typedef Matrix_<float, 2> Vector2f;
typedef Matrix_<float, 3> Vector3f;
typedef Matrix_<float, 4> Vector4f;
typedef Matrix_<double, 3> Vector3d;
typedef Matrix_<double, 4> Vector4d;

}  // namespace Eigen
#endif  // !HPMVS_HAVE_EIGEN
#endif  // HPMVS_VEC_H_
