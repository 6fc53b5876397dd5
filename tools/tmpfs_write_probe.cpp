// How fast one file in /dev/shm takes 1 GB: one write() stream, T pwrite() streams on disjoint ranges, T threads copying into an mmap of the file.
// usage: g++ -O2 tools/tmpfs_write_probe.cpp -o /tmp/pw -lpthread && /tmp/pw 16
#include <fcntl.h>
#include <unistd.h>
#include <sys/mman.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <thread>
#include <vector>
#include <time.h>
static double now(){struct timespec t;clock_gettime(CLOCK_MONOTONIC,&t);return t.tv_sec*1e3+t.tv_nsec*1e-6;}
int main(int argc,char**argv){
  const size_t N=(size_t)1<<30; int T=atoi(argv[1]); const char* path=argc>2?argv[2]:"/dev/shm/pw_test.bin";
  char* src=(char*)malloc(N); memset(src,'x',N);
  for(int mode=0;mode<3;mode++){
    unlink(path); int fd=open(path,O_CREAT|O_RDWR|O_TRUNC,0644);
    double t0=now();
    if(mode==0){ size_t off=0; while(off<N){ssize_t k=write(fd,src+off,std::min<size_t>(N-off,64<<20)); off+=k;} }
    else if(mode==1){ std::vector<std::thread> th; size_t per=N/T; for(int t=0;t<T;t++) th.emplace_back([=]{ size_t off=t*per,end=off+per; while(off<end){ssize_t k=pwrite(fd,src+off,std::min<size_t>(end-off,8<<20),off); off+=k;} }); for(auto&t:th)t.join(); }
    else { if(ftruncate(fd,N)) return 1; char* m=(char*)mmap(0,N,PROT_READ|PROT_WRITE,MAP_SHARED,fd,0); std::vector<std::thread> th; size_t per=N/T; for(int t=0;t<T;t++) th.emplace_back([=]{ memcpy(m+t*per,src+t*per,per); }); for(auto&t:th)t.join(); munmap(m,N); }
    close(fd);
    printf("%s: %.0f ms (%.1f GB/s)\n", mode==0?"write x1":mode==1?"pwrite xT":"mmap xT", now()-t0, N/1e6/(now()-t0));
  }
  unlink(path);
}
