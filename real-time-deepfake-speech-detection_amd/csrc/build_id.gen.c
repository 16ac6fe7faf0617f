const char afx_build_id_str[] = "baf4171166b0";
