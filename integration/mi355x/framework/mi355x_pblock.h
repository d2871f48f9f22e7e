// framework/core/mi355x_pblock.h — PBlock<MI355X>: the weight container of the framework (framework/core/parameter.h:192+)
// for the MI355X target: a device tensor (Tensor<MI355X>, hipMalloc'ed through TargetWrapper<MI355X>) plus its host mirror
// (Tensor<X86>), shared by reference between copies like every other PBlock specialisation. Included from parameter.h
// under USE_MI355X_PLACE (docs/Manual/addCustomDevice.md:330-345). The member list is the one the framework's operators
// call (d_tensor / h_tensor / vector / shape / real_shape / map_to_host / map_to_device / share_from ...).
#ifndef ANAKIN_FRAMEWORK_CORE_MI355X_PBLOCK_H
#define ANAKIN_FRAMEWORK_CORE_MI355X_PBLOCK_H

namespace anakin {

template <>
class PBlock<MI355X> {
public:
    typedef Tensor4d<MI355X> d_type;
    typedef Tensor4d<X86> h_type;

    PBlock(DataType dtype = AK_FLOAT) : _dev(std::make_shared<d_type>(dtype)), _host(std::make_shared<h_type>(dtype)) {}
    PBlock(Shape4d& shape, DataType dtype = AK_FLOAT)
        : _dev(std::make_shared<d_type>(shape, dtype)), _host(std::make_shared<h_type>(shape, dtype)) {}
    PBlock(const PBlock<MI355X>& o) = default;            // shallow: both copies refer to the same two tensors
    PBlock(PBlock<MI355X>& o) : PBlock(static_cast<const PBlock<MI355X>&>(o)) {}
    PBlock<MI355X>& operator=(const PBlock<MI355X>& o) = default;
    ~PBlock() {}

    inline bool host_only() { return false; }

    // device -> host mirror, whole allocation (an operator may have re-laid the device weights out)
    inline void map_to_host() { mirror(*_dev, *_host); }
    inline void map_to_device() { mirror(*_host, *_dev); }

    void share_from(const PBlock<MI355X>& o, const std::string& target) {
        *this = o;
        _shared = true;
        _share_from = target;
    }
    bool is_shared() { return _shared; }
    std::string share_target() { return _share_from; }

    d_type& d_tensor() { return *_dev; }
    h_type& h_tensor() { return *_host; }

    std::vector<float> vector() {
        const float* p = static_cast<const float*>(_host->data());
        return std::vector<float>(p, p + _host->valid_size());
    }

    void re_alloc(Shape4d shape) {
        _dev->re_alloc(shape);
        _host->re_alloc(shape);
    }
    Shape4d shape() const {
        CHECK(_dev->valid_shape() == _host->valid_shape()) << "PBlock<MI355X>: device and host shapes differ";
        return _dev->valid_shape();
    }
    DataType data_type() { return _host->get_dtype(); }
    Shape4d real_shape() { return _dev->shape(); }
    size_t count() const { return this->shape().count(); }

private:
    template <typename Src, typename Dst>
    static void mirror(Src& src, Dst& dst) {
        if (dst.get_dtype() != src.get_dtype()) dst.set_dtype(src.get_dtype());
        const Shape4d valid = src.valid_shape();
        const Shape4d real = src.shape();
        dst.re_alloc(real, dst.get_dtype());
        src.set_shape(real);
        dst.copy_from(src);
        src.set_shape(valid);
        dst.set_shape(valid);
    }

    std::shared_ptr<d_type> _dev;
    std::shared_ptr<h_type> _host;
    bool _shared{false};
    std::string _share_from;
};

}  // namespace anakin
#endif
