// Host model: the hierarchy with its fixed shape (leaf ranges halved, spheres sorted along the longest axis) against shapes chosen by surface area -- any cut / cuts at whole leaves -- on the camera, bounce and shadow rays of a scene: pair steps and leaf visits per ray and per wavefront.  g++ -O2 -o /tmp/sim_tree_shape tools/sim_tree_shape.cpp && python tools/sim_wide_nodes.py c3 | /tmp/sim_tree_shape
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>
struct S { float r, x, y, z; };
struct Box { float lo[3], hi[3]; };
struct Node { Box b; int left, right, first, count; };   // leaf if left < 0
static std::vector<S> sph;
static std::vector<Node> nodes;
static Box box_of(int first, int count) {
    Box b{{1e30f,1e30f,1e30f},{-1e30f,-1e30f,-1e30f}};
    for (int i = first; i < first + count; ++i) { const float c[3] = {sph[i].x, sph[i].y, sph[i].z};
        for (int k = 0; k < 3; ++k) { b.lo[k] = std::min(b.lo[k], c[k]-sph[i].r); b.hi[k] = std::max(b.hi[k], c[k]+sph[i].r); } }
    return b;
}
static float area(const Box &b) { float d[3] = {b.hi[0]-b.lo[0], b.hi[1]-b.lo[1], b.hi[2]-b.lo[2]}; return 2*(d[0]*d[1]+d[1]*d[2]+d[2]*d[0]); }
static int build(int first, int count, int mode) {
    Node n; n.b = box_of(first, count); n.left = n.right = -1; n.first = first; n.count = count;
    const int id = (int)nodes.size(); nodes.push_back(n);
    if (count <= 8) return id;
    int best_axis = 0, best_cut = count / 2; 
    if (mode == 0) {   // median by count along the longest axis of the centres' box, left gets whole leaves (as rt_bvh.hip)
        float lo[3] = {1e30f,1e30f,1e30f}, hi[3] = {-1e30f,-1e30f,-1e30f};
        for (int i = first; i < first+count; ++i) { const float c[3] = {sph[i].x, sph[i].y, sph[i].z}; for (int k=0;k<3;++k){lo[k]=std::min(lo[k],c[k]);hi[k]=std::max(hi[k],c[k]);} }
        for (int k = 1; k < 3; ++k) if (hi[k]-lo[k] > hi[best_axis]-lo[best_axis]) best_axis = k;
        const int leaves = (count + 7) / 8; best_cut = std::min(count, (leaves / 2) * 8);
        if (best_cut == 0) best_cut = count / 2;
    } else {
        float best = 1e30f;
        for (int ax = 0; ax < 3; ++ax) {
            std::sort(sph.begin()+first, sph.begin()+first+count, [ax](const S&p,const S&q){return (&p.x)[ax] < (&q.x)[ax];});
            std::vector<float> la(count), ra(count);
            Box b{{1e30f,1e30f,1e30f},{-1e30f,-1e30f,-1e30f}};
            for (int i = 0; i < count; ++i) { const S &s = sph[first+i]; const float c[3]={s.x,s.y,s.z}; for(int k=0;k<3;++k){b.lo[k]=std::min(b.lo[k],c[k]-s.r);b.hi[k]=std::max(b.hi[k],c[k]+s.r);} la[i]=area(b); }
            b = Box{{1e30f,1e30f,1e30f},{-1e30f,-1e30f,-1e30f}};
            for (int i = count-1; i >= 0; --i) { const S &s = sph[first+i]; const float c[3]={s.x,s.y,s.z}; for(int k=0;k<3;++k){b.lo[k]=std::min(b.lo[k],c[k]-s.r);b.hi[k]=std::max(b.hi[k],c[k]+s.r);} ra[i]=area(b); }
            const int step = mode == 2 ? 8 : 1;             // mode 2: cuts at multiples of 8 only (whole leaves on the left)
            for (int cut = step; cut < count; cut += step) {
                const float cost = la[cut-1]*std::ceil(cut/8.f) + ra[cut]*std::ceil((count-cut)/8.f);
                if (cost < best) { best = cost; best_axis = ax; best_cut = cut; }
            }
        }
    }
    const int ax = best_axis;
    std::sort(sph.begin()+first, sph.begin()+first+count, [ax](const S&p,const S&q){return (&p.x)[ax] < (&q.x)[ax];});
    const int l = build(first, best_cut, mode), r = build(first+best_cut, count-best_cut, mode);
    nodes[id].left = l; nodes[id].right = r;
    return id;
}
static bool hit_box(const Box &b, const float o[3], const float inv[3], float far, float &tn) {
    float t0 = 0.f, t1 = far;
    for (int k=0;k<3;++k){ float a=(b.lo[k]-o[k])*inv[k], c=(b.hi[k]-o[k])*inv[k]; if(a>c)std::swap(a,c); t0=std::max(t0,a); t1=std::min(t1,c);} tn=t0; return t0<=t1; }
static float hit_sphere(const S &s, const float o[3], const float d[3]) {
    const float op[3]={s.x-o[0],s.y-o[1],s.z-o[2]}; const float b=op[0]*d[0]+op[1]*d[1]+op[2]*d[2];
    float det=b*b-(op[0]*op[0]+op[1]*op[1]+op[2]*op[2])+s.r*s.r; if(det<0)return 0; det=std::sqrt(det);
    const float t1=b-det,t2=b+det; return t1>0.01f?t1:(t2>0.01f?t2:0.f); }
struct Counts { long steps=0, leaves=0, tests=0; };
static float walk(int root, const float o[3], const float d[3], float far, bool any, Counts &c) {
    float inv[3]; for(int k=0;k<3;++k) inv[k]=1.f/d[k];
    std::vector<int> st; st.push_back(root); bool done=false;
    while(!st.empty() && !done){ int id=st.back(); st.pop_back(); const Node &n=nodes[id];
        if(n.left<0){ c.leaves++; c.tests += 8; for(int i=n.first;i<n.first+n.count;++i){ float t=hit_sphere(sph[i],o,d); if(t>0&&t<far){far=t; if(any)done=true;} } continue; }
        c.steps++; float t0,t1; bool h0=hit_box(nodes[n.left].b,o,inv,far,t0), h1=hit_box(nodes[n.right].b,o,inv,far,t1);
        if(h0&&h1){ if(t0<t1){st.push_back(n.right);st.push_back(n.left);} else {st.push_back(n.left);st.push_back(n.right);} }
        else if(h0) st.push_back(n.left); else if(h1) st.push_back(n.right);
    }
    return far;
}
int main(){ int n; if(scanf("%d",&n)!=1)return 1; std::vector<S> all(n); for(auto&s:all) if(scanf("%f %f %f %f",&s.r,&s.x,&s.y,&s.z)!=4)return 1;
    float cam[12]; int w,h; for(float&v:cam) if(scanf("%f",&v)!=1)return 1; if(scanf("%d %d",&w,&h)!=2)return 1;
    std::vector<S> always, tree; std::vector<float> rr; for(auto&s:all) rr.push_back(std::fabs(s.r)); std::nth_element(rr.begin(),rr.begin()+n/2,rr.end());
    const float r_cut=8.f*rr[n/2]; for(auto&s:all)(std::fabs(s.r)<=r_cut?tree:always).push_back(s);
    struct Ray{float o[3],d[3],far;bool any;}; 
    for(int mode=0;mode<3;++mode){ sph=tree; nodes.clear(); const int root=build(0,(int)sph.size(),mode);
        int leaves=0; double sah=0; for(auto&nd:nodes){ if(nd.left<0)leaves++; sah+=area(nd.b);} 
        std::mt19937 rng(7); std::uniform_real_distribution<float> U(0.f,1.f);
        std::vector<Ray> prim,bounce,shadow; 
        for(int ty=0;ty<h;ty+=48) for(int tx=0;tx<w;tx+=48) for(int y=ty;y<ty+8;++y) for(int x=tx;x<tx+8;++x){ const float kx=(x+0.5f)/w-0.5f, ky=(y+0.5f)/h-0.5f; Ray r; float len=0; for(int k=0;k<3;++k){r.d[k]=cam[6+k]*kx+cam[9+k]*ky+cam[3+k]; len+=r.d[k]*r.d[k];} for(int k=0;k<3;++k){r.o[k]=cam[k]+0.1f*r.d[k]; r.d[k]/=std::sqrt(len);} r.far=1e20f;r.any=false; prim.push_back(r);} 
        auto run=[&](std::vector<Ray>&rays,const char*name,std::vector<float>*tout){ Counts tot; long ws=0,wl=0,waves=0; for(size_t base=0;base<rays.size();base+=64){ long ms=0,ml=0; for(size_t i=base;i<std::min(rays.size(),base+64);++i){ Counts c; float far=rays[i].far; for(auto&s:always){float t=hit_sphere(s,rays[i].o,rays[i].d); if(t>0&&t<far)far=t;} float t=walk(root,rays[i].o,rays[i].d,far,rays[i].any,c); if(tout)(*tout)[i]=t; tot.steps+=c.steps;tot.leaves+=c.leaves; ms=std::max(ms,c.steps); ml=std::max(ml,c.leaves);} ws+=ms;wl+=ml;waves++;} printf("  %-8s per ray %.2f steps %.2f leaves; per wavefront %.1f steps %.1f leaves\n",name,(double)tot.steps/rays.size(),(double)tot.leaves/rays.size(),(double)ws/waves,(double)wl/waves); };
        printf("mode %d (%s): %d leaves, %zu nodes, sum of node areas %.0f\n",mode,mode==0?"median by count, longest axis":mode==1?"SAH sweep, any cut":"SAH sweep, cuts at whole leaves",leaves,nodes.size(),sah);
        std::vector<float> t(prim.size()); run(prim,"primary",&t);
        const float light[3]={0.f,60.f,0.f};
        for(size_t i=0;i<prim.size();++i){ if(!(t[i]<1e19f))continue; float hp[3]; for(int k=0;k<3;++k)hp[k]=prim[i].o[k]+prim[i].d[k]*t[i]; float nrm[3]={0,1,0}; for(const S&q:sph){ float v[3]={hp[0]-q.x,hp[1]-q.y,hp[2]-q.z}; float dist=std::sqrt(v[0]*v[0]+v[1]*v[1]+v[2]*v[2]); if(std::fabs(dist-q.r)<1e-3f*q.r+1e-3f){for(int k=0;k<3;++k)nrm[k]=v[k]/dist;break;} }
            float uu[3]={nrm[2],0.f,-nrm[0]}; if(std::fabs(nrm[0])<0.1f&&std::fabs(nrm[2])<0.1f){uu[0]=0;uu[1]=-nrm[2];uu[2]=nrm[1];} float ul=std::sqrt(uu[0]*uu[0]+uu[1]*uu[1]+uu[2]*uu[2]); for(float&v:uu)v/=ul; float vv[3]={nrm[1]*uu[2]-nrm[2]*uu[1],nrm[2]*uu[0]-nrm[0]*uu[2],nrm[0]*uu[1]-nrm[1]*uu[0]};
            float r1=6.2831853f*U(rng),r2=U(rng),r2s=std::sqrt(r2),cz=std::sqrt(1-r2); Ray b; for(int k=0;k<3;++k){b.d[k]=uu[k]*std::cos(r1)*r2s+vv[k]*std::sin(r1)*r2s+nrm[k]*cz; b.o[k]=hp[k]+0.02f*nrm[k];} b.far=1e20f;b.any=false; bounce.push_back(b);
            Ray s; float len=0; for(int k=0;k<3;++k){s.d[k]=light[k]-hp[k];len+=s.d[k]*s.d[k];} len=std::sqrt(len); for(int k=0;k<3;++k){s.d[k]/=len;s.o[k]=hp[k]+0.02f*s.d[k];} s.far=len-7.f;s.any=true; shadow.push_back(s);} 
        run(bounce,"bounce",nullptr); run(shadow,"shadow",nullptr);
    } return 0; }
